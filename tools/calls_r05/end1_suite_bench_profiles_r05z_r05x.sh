#!/bin/bash
# round 5 end: full GPU suite, the default bench line with rocprofv3 stats + PMC passes (r05z), the round-5 extra set (r05x)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r05end; mkdir -p $o
timeout 1500 python -m pytest tests -m gpu -q --timeout 700 -p no:cacheprovider > $o/tests.log 2>&1; echo "tests rc=$?" | tee -a $o/tests.log
grep -n "passed\|failed\|^FAILED" $o/tests.log | tail -5
PMC_MIN=1 bash tools/collect_profiles.sh r05z > $o/collect.log 2>&1; tail -2 $o/collect.log
R5=1 bash tools/collect_profiles_extra.sh r05x > $o/collect_extra.log 2>&1; tail -2 $o/collect_extra.log
python __graft_entry__.py --smoke 2>&1 | tail -1

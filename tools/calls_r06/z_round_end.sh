#!/bin/bash
# round 6 end: full GPU suite, the default bench line with rocprofv3 stats + PMC passes (r06z), the round-6 extra set (r06x)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06end; mkdir -p $o
timeout 1500 python -m pytest tests -m gpu -q > $o/tests.log 2>&1; echo "rc=$?" >> $o/tests.log; tail -4 $o/tests.log | cut -c1-300
bash tools/collect_profiles.sh r06z > $o/collect.log 2>&1; tail -3 $o/collect.log
R6=1 bash tools/collect_profiles_extra.sh r06x > $o/collect_extra.log 2>&1; tail -3 $o/collect_extra.log
du -sh gpurun_out

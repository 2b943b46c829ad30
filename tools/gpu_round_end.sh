#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/s18; mkdir -p $o
timeout 2400 python -m pytest tests -m gpu -q > $o/tests.log 2>&1; echo "rc=$?" >> $o/tests.log; tail -4 $o/tests.log
bash tools/collect_profiles.sh r02d > $o/collect.log 2>&1; tail -2 $o/collect.log
bash tools/collect_profiles_extra.sh r02z > $o/collect_extra.log 2>&1; tail -3 $o/collect_extra.log

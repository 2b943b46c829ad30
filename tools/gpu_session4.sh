#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/s4; mkdir -p $o
timeout 1800 python -m pytest tests -m gpu -q --durations=6 > $o/tests.log 2>&1; echo "rc=$?" >> $o/tests.log; tail -12 $o/tests.log
bash tools/collect_profiles.sh r02b > $o/collect.log 2>&1; tail -2 $o/collect.log
bash tools/collect_profiles_extra.sh r02x > $o/collect_extra.log 2>&1; tail -3 $o/collect_extra.log

#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/s13; mkdir -p $o
bash tools/collect_profiles.sh r02c > $o/collect.log 2>&1; tail -2 $o/collect.log
bash tools/collect_profiles_extra.sh r02y > $o/collect_extra.log 2>&1; tail -3 $o/collect_extra.log
cat gpurun_out/r02c/bench.json | tail -1 | cut -c1-600

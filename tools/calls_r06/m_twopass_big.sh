#!/bin/bash
# round 6 call M: the two-pass path at ranking scale (configs[3] shard) and other shapes
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06m; mkdir -p $o
for args in "8192 1250000 256" "8192 1250000 64" "16384 1250000 256" "2048 1250000 256" "1024 300000 256" "100000 1250000 256"; do
  timeout 300 python tools/bench_score.py $args 2>&1 | grep -v amdgpu | tee -a $o/score.txt | cut -c1-200
done

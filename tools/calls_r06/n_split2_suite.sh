#!/bin/bash
# round 6 call N: second-chance split cap A/B (Q = 16384 x 1.25 M), then the whole suite (qna probe: are the encoders side by side?)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06n; mkdir -p $o
for m in 32 128; do
  for args in "16384 1250000 256" "8192 1250000 256" "4096 1250000 64"; do
    SSE_SPLIT2_MAX=$m timeout 300 python tools/bench_score.py $args 2>&1 | grep "^bf16  \|second chance" | sed "s/^/split2_max=$m: /" | cut -c1-200
  done
done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep "\[qna\] makefile\|passed\|failed" | cut -c1-600

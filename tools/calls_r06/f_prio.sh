#!/bin/bash
# round 6 call F: wave priorities of the gate-split kernel (GEMM phase, tail phase)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06f; mkdir -p $o
for lib in libsse_hip libsse_p02 libsse_p00 libsse_p03 libsse_p13; do
  SSE_HIP_LIB=$PWD/sequence-semantic-embedding_amd/$lib.so timeout 300 python tools/bench_shapes.py 50 96 64 80 40 64 50 50 50 128 64 80 > $o/$lib.txt 2>&1
  echo "== $lib"; grep "E=" $o/$lib.txt
done

#!/bin/bash
# round 6 call E: clock64 phase table of the gate-split kernel (measurement build libsse_gsclk.so) + A/B against lstm_fwd_kernel<2,1,1>
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06e; mkdir -p $o
SSE_HIP_LIB=$PWD/sequence-semantic-embedding_amd/libsse_gsclk.so timeout 300 python tools/bench_shapes.py 50 96 64 80 40 64 50 50 50 128 64 80 > $o/gsclk.txt 2>&1
cat $o/gsclk.txt | grep -v amdgpu.ids | cut -c1-330
for v in 0 1; do
  SSE_FWD_GS=$v timeout 300 python tools/bench_shapes.py > $o/shapes_gs$v.txt 2>&1
  echo "== SSE_FWD_GS=$v"; grep "E=" $o/shapes_gs$v.txt
done
timeout 600 python -m pytest tests/test_gpu_encode.py -x -q -k "gate_split" 2>&1 | tail -3

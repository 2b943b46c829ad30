#!/bin/bash
# clock64 phase table of the gate-split kernel (measurement build libsse_gsclk.so: -DSSE_GS_CLOCK)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06e; mkdir -p $o
SSE_HIP_LIB=$PWD/sequence-semantic-embedding_amd/libsse_gsclk.so timeout 300 python tools/bench_shapes.py 50 96 64 80 50 128 64 80 > $o/gsclk.txt 2>&1
cat $o/gsclk.txt | grep -v amdgpu.ids | cut -c1-330

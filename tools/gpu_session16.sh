#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
o=$GRAFT_REPO_ROOT/gpurun_out/s16; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_encode.py -m gpu -q -x -s -k "split_bf16" > $o/tests.log 2>&1; echo "rc=$?" >> $o/tests.log; grep -v amdgpu.ids $o/tests.log | tail -25
timeout 300 python tools/bench_x3.py 2>&1 | grep -v amdgpu.ids

#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
o=$GRAFT_REPO_ROOT/gpurun_out/s11; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_encode.py -m gpu -q -x -k "cluster or rows_independent" > $o/tests.log 2>&1; echo "rc=$?" >> $o/tests.log; tail -5 $o/tests.log
timeout 300 python tools/bench_query_encode.py 2>&1 | grep -v amdgpu.ids; timeout 300 python tools/bench_query_encode.py 256 256 2 2>&1 | grep -v amdgpu.ids; timeout 300 python tools/bench_query_encode.py 96 64 80 2>&1 | grep -v amdgpu.ids

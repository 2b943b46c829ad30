#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
o=$GRAFT_REPO_ROOT/gpurun_out/s20; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_fullsize.py -m gpu -q -x -k "train" > $o/tests.log 2>&1; echo "rc=$?" >> $o/tests.log; tail -3 $o/tests.log
timeout 300 python tools/bench_train.py 128 1024 8192 2>&1 | grep -v amdgpu.ids
SSE_TRAIN_REALISTIC=1 timeout 300 python tools/bench_train.py 8192 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/bench_train_default.py 2>&1 | grep -v amdgpu.ids

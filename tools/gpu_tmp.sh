#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -m gpu -q -k "train or fuzz" 2>&1 | grep -v amdgpu | tail -12
for x in 1 0; do echo "== train_bwd_x3=$x"; SSE_TRAIN_BWD_X3=$x timeout 300 python tools/bench_train.py 128 1024 8192 2>&1 | grep -v amdgpu.ids; done
timeout 300 python tools/bench_train_default.py 2>&1 | grep -v amdgpu.ids

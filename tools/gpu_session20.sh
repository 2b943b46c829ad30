#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
o=$GRAFT_REPO_ROOT/gpurun_out/s20; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_fullsize.py tests/test_gpu_cli.py tests/test_gpu_fuzz.py -m gpu -q -x -k "train or cli or fuzz" > $o/tests.log 2>&1; echo "rc=$?" >> $o/tests.log; tail -12 $o/tests.log
for x in 1 0; do echo "== train_dk_x3=$x"; SSE_TRAIN_DK_X3=$x timeout 300 python tools/bench_train.py 128 1024 8192 2>&1 | grep -v amdgpu.ids; done
timeout 300 python tools/bench_train_default.py 2>&1 | grep -v amdgpu.ids

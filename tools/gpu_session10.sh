#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
o=$GRAFT_REPO_ROOT/gpurun_out/s10; mkdir -p $o
timeout 1200 python -m pytest tests/test_gpu_encode.py tests/test_gpu_train.py tests/test_gpu_fullsize.py tests/test_gpu_cli.py -m gpu -q -x > $o/tests.log 2>&1; echo "rc=$?" >> $o/tests.log; tail -8 $o/tests.log
timeout 300 python tools/bench_train_default.py > $o/tdef.txt 2>&1; grep -v amdgpu.ids $o/tdef.txt
timeout 300 python tools/bench_configs.py > $o/configs.txt 2>&1; grep -v amdgpu.ids $o/configs.txt | tail -30

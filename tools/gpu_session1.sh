#!/bin/bash
# First GPU session of round 2: parity suite on the new kernels, then A/B bench (round-1 build vs this build).
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s1
python -c "import torch; print(torch.cuda.get_device_name(0))" > gpurun_out/s1/dev.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_fullsize.py > gpurun_out/s1/tests_main.log 2>&1
echo "rc=$?" >> gpurun_out/s1/tests_main.log
tail -5 gpurun_out/s1/tests_main.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --durations=12 > gpurun_out/s1/tests_full.log 2>&1
echo "rc=$?" >> gpurun_out/s1/tests_full.log
tail -25 gpurun_out/s1/tests_full.log
SSE_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libsse_hip_r01.so timeout 300 python bench.py --no-cpu-baseline > gpurun_out/s1/bench_r01.json 2> gpurun_out/s1/bench_r01.err
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/s1/bench_new.json 2> gpurun_out/s1/bench_new.err
tail -c 1500 gpurun_out/s1/bench_r01.json; echo; tail -c 1500 gpurun_out/s1/bench_new.json; tail -3 gpurun_out/s1/bench_new.err

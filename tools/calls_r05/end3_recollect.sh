#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r05end; mkdir -p $o
timeout 300 python -m pytest tests/test_gpu_cnn.py tests/test_gpu_fullsize.py tests/test_gpu_cli.py tests/test_gpu_generic.py -m gpu -q --timeout 200 -p no:cacheprovider -k "cnn or generic" > $o/tests_cnn.log 2>&1; echo "tests rc=$?"; grep -n "passed\|failed\|^FAILED" $o/tests_cnn.log | tail -3
timeout 120 python tools/bench_cnn.py 2>&1 | tail -6
PMC_MIN=1 timeout 900 bash tools/collect_profiles.sh r05z > $o/collect.log 2>&1; tail -2 $o/collect.log
R5=1 timeout 900 bash tools/collect_profiles_extra.sh r05x > $o/collect_extra.log 2>&1; tail -2 $o/collect_extra.log
ls gpurun_out/r05z gpurun_out/r05x | head -40

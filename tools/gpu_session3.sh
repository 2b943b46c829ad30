#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/s3; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_encode.py tests/test_gpu_fullsize.py -m gpu -q -k "encode or qna" > $o/tests.log 2>&1; echo "rc=$?" >> $o/tests.log; tail -4 $o/tests.log
python tools/bench_configs.py shapes c2 2>/dev/null | tee $o/shapes.txt

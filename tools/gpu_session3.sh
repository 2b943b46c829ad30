#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/s3; mkdir -p $o

timeout 900 python -m pytest tests/test_gpu_encode.py -m gpu -q -s > $o/tests.log 2>&1; echo "rc=$?" >> $o/tests.log; grep -n "few-sequences kernel vs\|passed\|failed\|Error\|^E " $o/tests.log | head -30
timeout 600 python tools/bench_demo_query.py 10000000 > $o/demo.log 2>&1; tail -16 $o/demo.log

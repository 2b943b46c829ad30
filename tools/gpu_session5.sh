#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/s5; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_cnn.py tests/test_gpu_train.py -m gpu -q -x > $o/tests.log 2>&1; echo "rc=$?" >> $o/tests.log; tail -8 $o/tests.log
timeout 300 python tools/bench_cnn.py > $o/cnn.txt 2>&1; tail -6 $o/cnn.txt
for r in 32 64 0; do
  echo "== lstm_train_rows=$r" >> $o/train.txt
  SSE_TRAIN_ROWS=$r timeout 300 python tools/bench_train.py 8192 >> $o/train.txt 2>&1
done
echo "== serial rows=64" >> $o/train.txt
SSE_TRAIN_SERIAL=1 SSE_TRAIN_ROWS=64 timeout 300 python tools/bench_train.py 8192 >> $o/train.txt 2>&1
cat $o/train.txt

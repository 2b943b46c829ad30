#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/s6; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_fullsize.py -m gpu -q -x -k "train" > $o/tests.log 2>&1; echo "rc=$?" >> $o/tests.log; tail -15 $o/tests.log
for d in 1 0; do
  echo "== train_pair_dedup=$d" >> $o/train.txt
  SSE_TRAIN_PAIR_DEDUP=$d timeout 300 python tools/bench_train.py 128 1024 8192 >> $o/train.txt 2>&1
done
SSE_TRAIN_SERIAL=1 timeout 300 python tools/bench_train.py 8192 >> $o/train.txt 2>&1
timeout 300 python tools/bench_train_default.py >> $o/train.txt 2>&1
cat $o/train.txt

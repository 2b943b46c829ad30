#!/bin/bash
# round 6 call B: device PAD-prefix bucketing test + qna recipe learning-rate exploration for the learned-model parity test
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06b; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_encode.py -x -q -k "device_pad_prefix or pad_prefix_skip" 2>&1 | tail -15
for lr in 0.02 0.05 0.1; do
  timeout 300 python tools/train_recipe_from_ids.py qna --epochs 40 --lr $lr --eval-every 10 > $o/qna_lr$lr.txt 2>&1
  grep -h "task specific\|oracle on\|trained " $o/qna_lr$lr.txt
done
timeout 300 python bench.py --no-scoring-leg --no-train-leg --no-x3-leg --no-cnn-leg --no-shapes-leg --no-sweep-leg --no-cpu-baseline --steps 5 --warmup 2 > $o/bench_rd.out 2>$o/bench_rd.err
tail -n 1 $o/bench_rd.out

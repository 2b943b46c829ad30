#!/bin/bash
# round 6 call C: bucket order reversed (most work first): A/B of pad_sort_dev on the real rows + the headline; new tests; 2 ranks on one GPU
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06c; mkdir -p $o
for v in 0 1 0 1; do
  SSE_PAD_SORT_DEV=$v timeout 300 python bench.py --no-scoring-leg --no-train-leg --no-x3-leg --no-cnn-leg --no-shapes-leg --no-sweep-leg --no-cpu-baseline --steps 20 --warmup 3 > $o/bench_rd$v.out 2>$o/bench_rd.err
  python - <<PY
import json
d=json.loads(open('$o/bench_rd$v.out').read().strip().splitlines()[-1]); r=d['legs']['realdata_c3']
full=json.load(open('profiles/bench_legs_latest.json'))['realdata_leg']
print('pad_sort_dev=$v', 'main kernel ms', d['roofline']['avg_kernel_ms'], 'step', d['ms_per_step'], {k:v for k,v in r.items() if 'pad_skip_1' in k}, 'host buffers', full['pad_skip_1']['index_build'].get('host_buffers_ms'), full['pad_skip_1']['query_encode'].get('host_buffers_ms'))
PY
done
timeout 600 python -m pytest tests/test_gpu_encode.py tests/test_gpu_rccl.py -x -q 2>&1 | tail -8
timeout 200 python tools/rccl_two_ranks_one_gpu.py > $o/two_ranks.txt 2>&1; echo "two ranks rc=$?"; tail -12 $o/two_ranks.txt
for lr in 0.005 0.01; do
  timeout 300 python tools/train_recipe_from_ids.py qna --epochs 120 --lr $lr --eval-every 20 > $o/qna_lr$lr.txt 2>&1
  grep -h "task specific\|oracle on\|trained " $o/qna_lr$lr.txt
done

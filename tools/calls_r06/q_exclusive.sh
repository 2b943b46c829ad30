#!/bin/bash
# round 6 call Q: one workgroup per CU for length-sorted padded batches (A/B: SSE_FWD_SHARED_CU=1 is the round-5 placement)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06q; mkdir -p $o
for v in shared exclusive shared exclusive; do
  if [ $v = shared ]; then export SSE_FWD_SHARED_CU=1; else unset SSE_FWD_SHARED_CU; fi
  timeout 300 python bench.py --no-scoring-leg --no-train-leg --no-x3-leg --no-cnn-leg --no-shapes-leg --no-sweep-leg --no-cpu-baseline --steps 20 --warmup 3 > $o/bench_$v.out 2>$o/bench.err
  tail -n 1 $o/bench_$v.out | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['legs']['realdata_c3']; print('$v', 'index build', r['index_build_pad_skip_1'], 'queries', r['query_encode_pad_skip_1'], 'job', r['whole_job_ms'], r['top1_equal_vs_oracle'])"
done
unset SSE_FWD_SHARED_CU
timeout 900 python -m pytest tests/test_gpu_encode.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3

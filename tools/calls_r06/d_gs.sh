#!/bin/bash
# round 6 call D: gate-split small-cell kernel: bit-identity tests, then the recipe-shape leg A/B (SSE_FWD_GS=0 / 1)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06d; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_encode.py -x -q -k "gate_split or small_cells or device_pad_prefix" 2>&1 | tail -15
for v in 0 1; do
  SSE_FWD_GS=$v timeout 300 python tools/bench_shapes.py > $o/shapes_gs$v.txt 2>&1
  echo "== SSE_FWD_GS=$v"; tail -12 $o/shapes_gs$v.txt
done
SSE_PAD_SORT_DEV=1 timeout 300 python bench.py --no-scoring-leg --no-train-leg --no-x3-leg --no-cnn-leg --no-shapes-leg --no-sweep-leg --no-cpu-baseline --steps 20 --warmup 3 > $o/bench_rd1.out 2>$o/bench_rd.err
tail -n 1 $o/bench_rd1.out | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('main kernel ms', d['roofline']['avg_kernel_ms'], d['legs']['realdata_c3'])"
NCCL_DEBUG=WARN timeout 200 python tools/rccl_two_ranks_one_gpu.py 2>&1 | grep -i "warn\|duplicate\|exit" | head -8

#!/bin/bash
# round 6 call P: pad_lead (thread per row) + wave-aggregated scatter: tests, kernel times (rocprofv3), real-data legs
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06p; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_encode.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -4
timeout 300 python bench.py --no-scoring-leg --no-train-leg --no-x3-leg --no-cnn-leg --no-shapes-leg --no-sweep-leg --no-cpu-baseline --steps 20 --warmup 3 > $o/bench_rd.out 2>$o/bench_rd.err
tail -n 1 $o/bench_rd.out | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('main kernel ms', d['roofline']['avg_kernel_ms'], d['legs']['realdata_c3'])"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$o/stats -o p -- python $OLDPWD/tools/bench_shapes.py 50 96 64 80 > $OLDPWD/$o/stats.log 2>&1 )
grep "pad_\|lstm_fwd_gs" $o/stats/p_kernel_stats.csv | cut -c1-200

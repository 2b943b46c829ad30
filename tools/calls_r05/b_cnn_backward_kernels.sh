#!/bin/bash
# round 5, call B: CNN backward kernels (MFMA dX, pipelined dW): parity tests, timings, per-kernel profile
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r05b; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_cnn.py tests/test_gpu_fullsize.py tests/test_gpu_trained_parity.py tests/test_gpu_cli.py -m gpu -q --timeout 600 -p no:cacheprovider -k "cnn or crosslingual or standin" > $o/tests.log 2>&1; echo "tests rc=$?"
grep -n "passed\|failed\|^FAILED\|^\[cross" $o/tests.log | tail -12
python tools/bench_cnn.py 2>&1 | tail -6 | tee $o/cnn.txt
SSE_CNN_DX_GATHER=1 python tools/bench_cnn.py 2>&1 | tail -2 | tee $o/cnn_gather_dx.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$o/prof -o p -- python $GRAFT_REPO_ROOT/tools/bench_cnn.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r05b/prof/p_kernel_stats.csv')))
for r in rows[:22]: print(r['Name'][:90], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'], r['Percentage'])
PY
python tools/bench_c3.py 2>&1 | tail -6 | tee $o/c3.txt
find $o -name "*.csv" -size +5M -delete

#!/bin/bash
# quick dev loop: train tests, concurrent + serial timing, per-phase clocks of the BPTT kernel (measurement build)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
tag=$1; o=gpurun_out/$tag; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_fullsize.py -m gpu -q -x -k "train or parallel or packed" > $o/tests.log 2>&1; echo "rc=$?" >> $o/tests.log; tail -6 $o/tests.log
python tools/bench_train.py 128 1024 8192 2>&1 | grep B_rows
python tools/bench_train_default.py 2>&1 | grep B_rows
root=$(pwd)
( cd /tmp && SSE_TRAIN_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$o/serial -o p -- python $root/tools/bench_train.py 8192 > $root/$o/serial.log 2>&1 )
python - <<PY
import csv
rows=list(csv.DictReader(open('$o/serial/p_kernel_stats.csv')))
for r in rows[:8]: print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), ('%.1f us' % (float(r['AverageNs'])/1e3)).rjust(12), r['Percentage'])
PY
if [ -f sequence-semantic-embedding_amd/libsse_hip_clk.so ]; then
SSE_TRAIN_SERIAL=1 SSE_HIP_LIB=$PWD/sequence-semantic-embedding_amd/libsse_hip_clk.so python tools/bench_train.py 8192 2>&1 | grep "clock" | tail -8
fi

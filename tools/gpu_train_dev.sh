#!/bin/bash
# dev loop for the train step: train tests + timing + rocprof stats (serial and concurrent)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
tag=$1; o=gpurun_out/$tag; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_fullsize.py -m gpu -q -x -k "train or parallel or packed" > $o/tests.log 2>&1; echo "rc=$?" >> $o/tests.log; tail -25 $o/tests.log
root=$(pwd)
python tools/bench_train.py 128 1024 8192 > $o/train.txt 2>&1; cat $o/train.txt
python tools/bench_train_default.py > $o/train_default.txt 2>&1; cat $o/train_default.txt
( cd /tmp && SSE_TRAIN_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$o/serial -o p -- python $root/tools/bench_train.py 8192 > $root/$o/serial.log 2>&1 )
python - <<PY
import csv
rows=list(csv.DictReader(open('$o/serial/p_kernel_stats.csv')))
for r in rows[:12]: print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), ('%.1f us' % (float(r['AverageNs'])/1e3)).rjust(12), r['Percentage'])
PY

#!/bin/bash
# rocprofv3 kernel stats of the LSTM training step (tools/bench_train.py <rows>); run from the repo root on the GPU box
rows=${1:-8192}
root=$(pwd)
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/train$rows" -o p -- python "$root/tools/bench_train.py" $rows 2>&1 | grep B_rows
python - <<PY
import csv
rows=list(csv.DictReader(open('$root/gpurun_out/train$rows/p_kernel_stats.csv')))
for r in rows[:14]: print(r['Name'][:64].ljust(64), r['Calls'].rjust(5), ('%.1f us' % (float(r['AverageNs'])/1e3)).rjust(12), r['Percentage'])
PY

#!/bin/bash
# rocprofv3 kernel stats of the few-queries scoring path (tools/bench_demo_query.py); run from the repo root on the GPU box
root=$(pwd)
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/demo" -o p -- python "$root/tools/bench_demo_query.py" 2>&1 | grep "Q="
python - <<PY
import csv
rows=list(csv.DictReader(open('$root/gpurun_out/demo/p_kernel_stats.csv')))
for r in rows[:10]: print(r['Name'][:64].ljust(64), r['Calls'].rjust(5), ('%.1f us' % (float(r['AverageNs'])/1e3)).rjust(12), ('min %.1f' % (float(r['MinNs'])/1e3)), r['Percentage'])
PY

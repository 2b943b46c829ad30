#!/bin/bash
# One gpurun call for the single-query path: its tests, the parts of the call (tools/bench_latency.py), and a kernel trace
# of a few calls.  usage (GPU box, repo root): bash tools/gpu_latency.sh <tag>
export TMPDIR=/tmp
root=$(pwd)
tag=${1:-lat}; o=$root/gpurun_out/$tag; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_score.py tests/test_gpu_encode.py -m gpu -q -x > $o/tests.log 2>&1; echo "rc=$?" >> $o/tests.log; tail -5 $o/tests.log
timeout 300 python tools/bench_latency.py > $o/latency.txt 2>&1; cat $o/latency.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace -o p -- python $root/tools/bench_latency.py 1250000 11 > /dev/null 2>&1
python - <<PY
import csv, glob
f = glob.glob('$o/trace/**/p_kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
with open('$o/kernel_stats.txt', 'w') as out:
    for r in rows[:16]:
        line = "%-70s %5s %10.1f us  min %8.1f  max %8.1f" % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3)
        print(line); out.write(line + "\n")
PY

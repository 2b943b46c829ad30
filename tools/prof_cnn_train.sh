cd /root/repo
python tools/bench_cnn.py 2>&1 | tail -4
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/cnntrain -o p -- python /root/repo/tools/bench_cnn.py > /dev/null 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('/root/repo/gpurun_out/cnntrain/p_kernel_stats.csv')))
for r in rows[:16]: print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
PY

export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
for lib in libsse_hip.so libsse_hip_m1.so libsse_hip_m2.so libsse_hip_m3.so; do
  echo "== $lib"
  ( cd /tmp && SSE_TRAIN_SERIAL=1 SSE_HIP_LIB=$GRAFT_REPO_ROOT/sequence-semantic-embedding_amd/$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$lib -o p -- python $GRAFT_REPO_ROOT/tools/bench_train.py 8192 2>&1 | grep B_rows )
  python - <<PY
import csv
rows=list(csv.DictReader(open('/tmp/ab_$lib/p_kernel_stats.csv')))
for r in rows[:2]: print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), ('%.1f us' % (float(r['AverageNs'])/1e3)).rjust(12))
PY
done

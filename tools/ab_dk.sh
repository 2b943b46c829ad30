export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_fullsize.py -m gpu -q -x -k "oracle or generations or paired or c2_lstm" 2>&1 | tail -3
for v in THREE TWO; do
  echo "== $v"
  ( cd /tmp && if [ $v = TWO ]; then export SSE_DK_TWO=1; fi; SSE_TRAIN_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abdk_$v -o p -- python $GRAFT_REPO_ROOT/tools/bench_train.py 8192 2>&1 | grep B_rows )
  python - <<PY
import csv
rows=list(csv.DictReader(open('/tmp/abdk_$v/p_kernel_stats.csv')))
for r in rows[:5]: print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), ('%.1f us' % (float(r['AverageNs'])/1e3)).rjust(12))
PY
done
python tools/bench_train.py 128 8192 2>&1 | grep B_rows
python tools/bench_train_default.py 2>&1 | grep B_rows

#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r05e; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_cnn.py tests/test_gpu_fullsize.py tests/test_gpu_cli.py -m gpu -q --timeout 300 -p no:cacheprovider -k "cnn" > $o/tests.log 2>&1; echo "tests rc=$?"
grep -n "passed\|failed\|^FAILED" $o/tests.log | tail -5
python tools/bench_cnn.py 2>&1 | tail -6 | tee $o/cnn.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$o/prof -o p -- python $GRAFT_REPO_ROOT/tools/bench_cnn.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r05e/prof/p_kernel_stats.csv')))
for r in rows[:14]: print(r['Name'][:90], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'], r['Percentage'])
PY
find $o -name "*.csv" -size +5M -delete

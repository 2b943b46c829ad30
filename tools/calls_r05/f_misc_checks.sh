#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r05f; mkdir -p $o
for mt in 256 200 120 96 60; do echo "== SSE_SPLIT_MIN_TILES=$mt"; SSE_SPLIT_MIN_TILES=$mt python tools/bench_c3.py 2>&1 | grep -v "amdgpu.ids\|top-1 score"; done | tee $o/c3_sweep.txt
cd /tmp
SSE_SPLIT_MIN_TILES=96 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$o/prof -o p -- python $GRAFT_REPO_ROOT/tools/bench_c3.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r05f/prof/p_kernel_stats.csv')))
for r in rows[:12]: print(r['Name'][:100], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'], r['Percentage'])
PY
find $o -name "*.csv" -size +5M -delete

#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r05rows; mkdir -p $o
for r in 64 32; do
  SSE_FWD_ROWS=$r timeout 120 python bench.py --no-scoring-leg --no-train-leg --no-x3-leg --no-cnn-leg --no-shapes-leg --no-sweep-leg --no-cpu-baseline --steps 3 --warmup 1 > $o/b$r.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('$o/b$r.json').read().strip().splitlines()[-1]); r=d['realdata_leg']
print('SSE_FWD_ROWS=$r', {k:(round(r[k]['index_build']['encode_ms'],3), round(r[k]['query_encode']['encode_ms'],3), round(r[k]['index_build']['host_buffers_ms'],2), round(r[k]['query_encode']['host_buffers_ms'],2)) for k in ('pad_skip_0','pad_skip_1')}, 'main', round(d['ms_per_step'],3))
PY
done

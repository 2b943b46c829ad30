#!/bin/bash
# one call: tile-rows policy check (no SSE_FWD_ROWS), the GPU suite, and the r05z evidence re-collected on the same sources
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r05rows; mkdir -p $o
timeout 120 python bench.py --no-scoring-leg --no-train-leg --no-x3-leg --no-cnn-leg --no-shapes-leg --no-sweep-leg --no-cpu-baseline --steps 3 --warmup 1 > $o/bauto.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('$o/bauto.json').read().strip().splitlines()[-1]); r=d['realdata_leg']
print('auto', {k:(round(r[k]['index_build']['encode_ms'],3), round(r[k]['query_encode']['encode_ms'],3)) for k in ('pad_skip_0','pad_skip_1')}, 'main', round(d['ms_per_step'],3))
PY
timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
PMC_MIN=1 bash tools/collect_profiles.sh r05z 2>&1 | tail -5

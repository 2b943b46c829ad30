#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/s3; mkdir -p $o
for v in 1 3 5 7; do
  SSE_SCORE_DBG=$v python bench.py --no-cpu-baseline --no-train-leg > $o/b_dbg$v.json 2>/dev/null
  python - $o/b_dbg$v.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "bf16 %.3f ms | fp32 %.3f ms" % (d["scoring_leg"]["ms_per_pass"], d["scoring_leg_fp32_candidates"]["ms_per_pass"]))
PY
done

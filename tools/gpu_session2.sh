#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/s2; mkdir -p $o
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $o/tests.log 2>&1; echo "rc=$?" >> $o/tests.log; tail -14 $o/tests.log
python bench.py --no-cpu-baseline > $o/b_default.json 2>$o/b_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/s2/b_default.json").read().strip().splitlines()[-1])
print("value %.0f enc_ms %.4f frac %.4f | bf16 %.3f ms | fp32 %.3f ms same=%s | train %.3f ms" % (d["value"], d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"], d["scoring_leg"]["ms_per_pass"], d["scoring_leg_fp32_candidates"]["ms_per_pass"], d["scoring_leg_fp32_candidates"]["identical_to_default"], d["train_leg"]["ms_per_step"]))
PY

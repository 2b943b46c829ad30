#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/s2; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_score.py tests/test_gpu_fullsize.py -m gpu -q -x -k "not train and not crosslingual" > $o/tests.log 2>&1; echo "rc=$?" >> $o/tests.log; tail -4 $o/tests.log
python bench.py --no-cpu-baseline --no-train-leg > $o/b_default.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open("gpurun_out/s2/b_default.json").read().strip().splitlines()[-1])
print("enc_ms %.4f frac %.4f | bf16 %.3f ms | fp32 %.3f ms same=%s" % (d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"], d["scoring_leg"]["ms_per_pass"], d["scoring_leg_fp32_candidates"]["ms_per_pass"], d["scoring_leg_fp32_candidates"]["identical_to_default"]))
PY

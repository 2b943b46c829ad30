"""bench.py driver contract on the GPU: one JSON object as the LAST line of stdout (also when RCCL prints its version
banner through C stdio), required keys present, and every multi-GPU branch exercised in a 1-rank RCCL group
(SSE_BENCH_FORCE_DIST=1)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("force_dist", ["0", "1"])
def test_bench_prints_one_json_line_last(force_dist):
    env = dict(os.environ, SSE_BENCH_FORCE_DIST=force_dist, NCCL_DEBUG="VERSION", MASTER_PORT="29547")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--batch", "2048", "--score-rows", "40000", "--score-queries", "512", "--score-iters", "1",
                          "--train-rows", "256", "--train-iters", "1", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines[-1]) < 8192, "headline line of %d bytes: the driver's parser lost a 23.5 KB line in round 5" % len(lines[-1])
    d = json.loads(lines[-1])
    assert sum(1 for l in lines if l.lstrip().startswith("{")) == 1
    legs_lines = [l for l in lines if l.startswith("BENCH_LEGS ")]
    assert len(legs_lines) == 1
    full = json.loads(legs_lines[0][len("BENCH_LEGS "):])
    assert full["scoring_leg"]["top1_planted_acc"] == 1.0 and full["value"] == pytest.approx(d["value"], rel=1e-5)
    assert json.load(open(os.path.join(ROOT, d["legs_file"])))["value"] == full["value"]
    assert d["rccl_ranks_seen"] == 1 and d["shard_bounds"] == [[0, 40000]]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0 and d["scaling"] == "weak"
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert d["config"]["workload"] and d["legs"]["score_bf16_sweep"]["planted_top1"] == 1.0
    assert d["legs"]["score_bf16_sweep"]["identical_to_fp32"] is True
    assert d["top1_match_vs_oracle"] == 1.0

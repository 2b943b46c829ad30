"""bench.py's evidence plumbing (no GPU): the PMC-derived roofline fields are quoted only while profiles/pmc_summary.json
describes the kernel sources in the tree (traffic used to come from a static file)."""
import json
import os

import bench


def test_csrc_sha_follows_the_kernel_sources(tmp_path, monkeypatch):
    root = tmp_path
    csrc = root / "sequence-semantic-embedding_amd" / "csrc"
    csrc.mkdir(parents=True)
    (csrc / "a.hip").write_text("kernel one\n")
    (csrc / "b.h").write_text("header\n")
    (csrc / "notes.txt").write_text("not a source\n")
    monkeypatch.setattr(bench, "ROOT", str(root))
    first = bench.csrc_sha()
    assert len(first) == 16 and first == bench.csrc_sha()
    (csrc / "notes.txt").write_text("still not a source\n")
    assert bench.csrc_sha() == first
    (csrc / "a.hip").write_text("kernel one, edited\n")
    assert bench.csrc_sha() != first


def test_pmc_summary_is_dropped_when_the_sources_changed(tmp_path, monkeypatch):
    root = tmp_path
    csrc = root / "sequence-semantic-embedding_amd" / "csrc"
    csrc.mkdir(parents=True)
    (csrc / "a.hip").write_text("kernel\n")
    (root / "profiles").mkdir()
    monkeypatch.setattr(bench, "ROOT", str(root))
    assert bench.pmc_summary() == {}                                     # nothing collected yet
    summary = {"csrc_sha": bench.csrc_sha(), "lstm_fwd_hbm_bytes_per_launch": 1.0, "mfma_busy": {"k": 0.5}}
    (root / "profiles" / "pmc_summary.json").write_text(json.dumps(summary))
    assert bench.pmc_summary() == summary
    (csrc / "a.hip").write_text("kernel, edited after the profile run\n")
    stale = bench.pmc_summary()
    assert set(stale) == {"stale"} and summary["csrc_sha"] in stale["stale"]
    assert stale.get("mfma_busy", {}).get("k") is None                   # what bench.py's .get() chains then report: null


def test_tracked_summary_matches_the_tree():
    """The committed summary should describe the committed sources (re-run tools/collect_profiles.sh after kernel edits)."""
    path = os.path.join(bench.ROOT, "profiles", "pmc_summary.json")
    if not os.path.exists(path):
        return
    d = json.load(open(path))
    assert "csrc_sha" in d and isinstance(d.get("mfma_busy", {}), dict)


def test_usable_cores_respects_affinity_and_quota():
    """The CPU baseline sizes its replicas by the cores the process may USE (the driver's container gets 16 of 256 hardware
    threads through a cgroup quota: replicas beyond it only thrash)."""
    n = bench.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1)
    if hasattr(os, "sched_getaffinity"):
        assert n <= len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            assert n <= max(1, int(int(quota) / int(period)))
    except (OSError, ValueError):
        pass


def test_busy_of_selects_kernels_by_name_prefix(monkeypatch):
    monkeypatch.setattr(bench, "PMC", {"mfma_busy": {"lstm_bwd2_kernel<8, true>": 0.77, "dk_gemm3_kernel<10, false>": 0.85,
                                                     "score_topk_kernel<4, true, false, true>": 0.69}})
    assert bench.busy_of("lstm_bwd", "dk_gemm") == {"lstm_bwd2_kernel<8, true>": 0.77, "dk_gemm3_kernel<10, false>": 0.85}
    assert bench.busy_of("cnn_") is None
    monkeypatch.setattr(bench, "PMC", {"stale": "sources changed"})
    assert bench.busy_of("lstm_bwd") is None


def test_headline_line_stays_small_and_keeps_the_contract_keys():
    """the 23.5 KB line of round 5 came back from the driver with `parsed: null`.  The headline is now a
    compact summary (full legs go to profiles/bench_legs_latest.json and an earlier BENCH_LEGS line); checked here on the
    round-5 full line, and on one with every leg inflated so the shedding guard is exercised."""
    full = json.load(open(os.path.join(bench.ROOT, "profiles", "r05z_bench.json")))
    text = bench.compact_headline(full)
    assert len(text) < bench.HEADLINE_MAX_BYTES < 8192 and "\n" not in text
    d = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "speedup_vs_cpu_baseline",
                "top1_match_vs_oracle", "legs"):
        assert key in d, key
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"])
    assert d["roofline"]["frac"] == float("%.6g" % full["roofline"]["frac"]) and d["config"] == full["config"]
    assert d["legs"]["train_fp32"]["ms"] > 0 and d["legs"]["c4_full_1gpu"]["planted_top1"] == 1.0
    fat = dict(full)
    fat["encode_leg_reference_shapes"] = dict(full["encode_leg_reference_shapes"],
                                              shapes=full["encode_leg_reference_shapes"]["shapes"] * 60)
    text = bench.compact_headline(fat)
    d = json.loads(text)
    assert len(text) <= bench.HEADLINE_MAX_BYTES and d["legs_truncated"] and "roofline" in d and "cpu_baseline" in d

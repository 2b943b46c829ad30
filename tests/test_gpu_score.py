"""Parity of the HIP scoring + top-k path (through the C ABI) against the
reference's own scoring code (golden fixtures) and the CPU oracle.
Bar: top-k row ids bit-exact (ties: lower row first); scores within 1e-12 of
the float64 reference (they are computed in float64 on the device)."""
import os

import numpy as np
import pytest

from oracle import sse_oracle as O
from tests.util import make_pair, model_params, random_ids

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _scorer(S=8):
    params = model_params("dual-encoder", 50, 8, 16, 16, S, 4)
    m, _ = make_pair(params)
    return m.handle


def _unit(rng, n, s):
    x = rng.standard_normal((n, s)).astype(np.float32)
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


@pytest.mark.parametrize("name", ["small", "eval"])
def test_topk_matches_reference_golden(name):
    z = np.load(os.path.join(G, "scoring_%s.npz" % name))
    h = _scorer()
    h.index_upload(z["tgt64"])                       # float64 rows as Evaluator.__init__ parses them
    for k in (1, 3, 10, 16):
        k = min(k, z["tgt64"].shape[0])
        sc, ids = h.score_topk(z["src"], k)
        assert np.array_equal(ids, z["ranked_idx"][:, :k])
        assert np.abs(sc - z["ranked_score"][:, :k]).max() < 1e-12


@pytest.mark.parametrize("Q,N,S", [(300, 5000, 64), (1, 777, 50), (129, 33, 256), (600, 32060, 64), (5, 16, 8)])
def test_topk_matches_oracle_random(Q, N, S):
    rng = np.random.RandomState(Q + N)
    q, t = _unit(rng, Q, S), _unit(rng, N, S)
    h = _scorer()
    h.index_upload(t)
    k = min(10, N)
    sc, ids = h.score_topk(q, k)
    wsc, wids = O.topk(O.scores_f64(q, t.astype(np.float64)), k)
    assert np.array_equal(ids, wids)
    assert np.abs(sc - wsc).max() < 1e-12


def test_unnormalised_queries_demo_path():
    """sse_demo.py:123 scores with the UN-normalised source encoding."""
    rng = np.random.RandomState(5)
    q = (rng.standard_normal((7, 64)) * 37.0).astype(np.float32)
    t = _unit(rng, 4000, 64)
    h = _scorer()
    h.index_upload(t)
    sc, ids = h.score_topk(q, 10)
    wsc, wids = O.topk(O.scores_f64(q, t.astype(np.float64)), 10)
    assert np.array_equal(ids, wids)
    assert np.abs(sc - wsc).max() < 1e-9


def test_exact_ties_rank_lower_row_first():
    rng = np.random.RandomState(6)
    t = _unit(rng, 200, 32)
    t[150] = t[3]
    t[77] = t[3]
    t[199] = t[120]
    q = np.concatenate([t[3:4], t[120:121], _unit(rng, 30, 32)])
    h = _scorer()
    h.index_upload(t)
    sc, ids = h.score_topk(q, 5)
    wsc, wids = O.topk(O.scores_f64(q, t.astype(np.float64)), 5)
    assert ids[0, :3].tolist() == [3, 77, 150] and ids[1, :2].tolist() == [120, 199]
    assert np.array_equal(ids, wids)


def test_near_ties_need_float64_rescoring():
    """Rows that differ by less than fp32 resolution of the score: ordering must
    follow the float64 reference arithmetic, not the fp32 GEMM."""
    rng = np.random.RandomState(8)
    S, N = 64, 3000
    t = _unit(rng, N, S).astype(np.float64)
    base = t[10].copy()
    for j, r in enumerate((500, 20, 2500, 1234)):        # tiny, distinct perturbations
        t[r] = base * (1.0 - (j + 1) * 1e-9)
    q = base[None, :].astype(np.float32)
    h = _scorer()
    h.index_upload(t)
    sc, ids = h.score_topk(q, 5)
    wsc, wids = O.topk(O.scores_f64(q, t), 5)
    assert np.array_equal(ids, wids)
    assert np.abs(sc - wsc).max() < 1e-12


def test_adversarial_ascending_scores_and_argument_checks():
    import sse_amd
    S, N = 16, 4096
    base = np.zeros(S, np.float32)
    base[0] = 1
    t = np.tile(base, (N, 1)) * np.linspace(0.1, 1.0, N, dtype=np.float32)[:, None]   # every row beats the last
    h = _scorer()
    h.index_upload(t)
    sc, ids = h.score_topk(base[None], 10)
    assert ids[0].tolist() == list(range(N - 1, N - 11, -1))
    sc17, ids17 = h.score_topk(base[None], 17)      # beyond the fused kernel's list: exact paging path
    assert ids17[0].tolist() == list(range(N - 1, N - 18, -1))
    with pytest.raises(sse_amd.SSEError):
        h.score_topk(base[None], 0)


def test_sharded_index_equals_unsharded():
    """8 logical shards on one GPU, per-shard top-k with global ids,
    k-way merge == unsharded top-k exactly."""
    import torch
    rng = np.random.RandomState(11)
    Q, N, S, k, P = 257, 8000, 64, 10, 8
    q, t = _unit(rng, Q, S), _unit(rng, N, S)
    t[4000] = t[10]                                  # tie across shards
    h = _scorer()
    h.index_upload(t)
    want_s, want_i = h.score_topk(q, k)
    dev = torch.device("cuda:0")
    qd = torch.from_numpy(q).to(dev)
    all_s = torch.empty((P, Q, k), dtype=torch.float64, device=dev)
    all_i = torch.empty((P, Q, k), dtype=torch.int64, device=dev)
    bounds = np.linspace(0, N, P + 1).astype(int)
    for p in range(P):
        rows = torch.from_numpy(t[bounds[p]:bounds[p + 1]]).to(dev)
        h.index_set_dev(rows.data_ptr(), rows.shape[0], S, id_base=int(bounds[p]))
        h.score_topk_dev(qd.data_ptr(), Q, k, all_s[p].data_ptr(), all_i[p].data_ptr())
    out_s = torch.empty((Q, k), dtype=torch.float64, device=dev)
    out_i = torch.empty((Q, k), dtype=torch.int64, device=dev)
    h.merge_topk_dev(all_s.data_ptr(), all_i.data_ptr(), P, Q, k, out_s.data_ptr(), out_i.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(out_i.cpu().numpy(), want_i)
    assert np.array_equal(out_s.cpu().numpy(), want_s)
    wsc, wids = O.topk(O.scores_f64(q, t.astype(np.float64)), k)
    assert np.array_equal(want_i, wids)


def test_large_index_properties():
    """Ranking-scale shard (1.25M x 256 would be C4's per-GPU shard; 400k here keeps
    the test short): planted neighbours are found, scores sorted, ids unique."""
    import torch
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    N, S, Q, k = 400_000, 256, 1024, 10
    t = torch.nn.functional.normalize(torch.randn((N, S), generator=g, device=dev), dim=1)
    q = torch.nn.functional.normalize(torch.randn((Q, S), generator=g, device=dev), dim=1)
    planted = torch.randint(0, N, (Q,), generator=g, device=dev)
    planted = torch.unique(planted)[:Q]
    Qp = planted.numel()
    t[planted] = torch.nn.functional.normalize(q[:Qp] + 0.1 * torch.randn((Qp, S), generator=g, device=dev), dim=1)
    h = _scorer()
    h.index_set_dev(t.data_ptr(), N, S)
    out_s = torch.empty((Q, k), dtype=torch.float64, device=dev)
    out_i = torch.empty((Q, k), dtype=torch.int64, device=dev)
    h.score_topk_dev(q.data_ptr(), Q, k, out_s.data_ptr(), out_i.data_ptr())
    torch.cuda.synchronize()
    s, i = out_s.cpu().numpy(), out_i.cpu().numpy()
    assert np.array_equal(i[:Qp, 0], planted.cpu().numpy())
    assert np.all(np.diff(s, axis=1) <= 0)
    assert all(len(set(r)) == k for r in i)
    # exact check on a slice against the float64 oracle
    tq = q[:8].cpu().numpy()
    wsc, wids = O.topk(O.scores_f64(tq, t.cpu().numpy().astype(np.float64)), k)
    assert np.array_equal(i[:8], wids)
    assert np.abs(s[:8] - wsc).max() < 1e-12


def test_many_queries_round_balanced_splits():
    """Q = 20,000 (157 query blocks): the launch policy splits the index further so that the workgroups fill the
    256 CUs in whole rounds (sse_api.hip score_dev_locked).  Every query is independent, so a sample of them is
    checked exactly against the float64 oracle; all are checked for planted top-1 / sortedness / uniqueness."""
    import torch
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    N, S, Q, k = 70_000, 64, 20_000, 10
    t = torch.nn.functional.normalize(torch.randn((N, S), generator=g, device=dev), dim=1)
    q = torch.nn.functional.normalize(torch.randn((Q, S), generator=g, device=dev), dim=1)
    rows = torch.randperm(N, generator=g, device=dev)[:Q]
    t[rows] = torch.nn.functional.normalize(q + 0.05 * torch.randn((Q, S), generator=g, device=dev), dim=1)
    h = _scorer()
    h.index_set_dev(t.data_ptr(), N, S)
    out_s = torch.empty((Q, k), dtype=torch.float64, device=dev)
    out_i = torch.empty((Q, k), dtype=torch.int64, device=dev)
    h.score_topk_dev(q.data_ptr(), Q, k, out_s.data_ptr(), out_i.data_ptr())
    torch.cuda.synchronize()
    s, i = out_s.cpu().numpy(), out_i.cpu().numpy()
    assert np.array_equal(i[:, 0], rows.cpu().numpy())
    assert np.all(np.diff(s, axis=1) <= 0)
    sample = np.random.RandomState(0).choice(Q, 200, replace=False)
    wsc, wids = O.topk(O.scores_f64(q.cpu().numpy()[sample], t.cpu().numpy().astype(np.float64)), k)
    assert np.array_equal(i[sample], wids)
    assert np.abs(s[sample] - wsc).max() < 1e-12


@pytest.mark.parametrize("Q,N,S", [(1, 200_000, 64), (7, 150_000, 256), (40, 300_000, 64), (32, 5000, 50)])
def test_few_queries_many_splits_merged_lists(Q, N, S):
    """Demo / web regime (sse_demo.py:121-134): a handful of queries against a large index uses the
    single-query-tile kernel with up to 256 index splits and in-workgroup list merging."""
    import torch
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(Q + N)
    t = torch.nn.functional.normalize(torch.randn((N, S), generator=g, device=dev), dim=1)
    q = torch.nn.functional.normalize(torch.randn((Q, S), generator=g, device=dev), dim=1)
    t[N - 1] = q[0]                      # best match in the very last (partial) tile
    t[12345 % N] = q[0]                  # and an exact duplicate earlier: tie -> lower row first
    h = _scorer()
    h.index_set_dev(t.data_ptr(), N, S)
    out_s = torch.empty((Q, 10), dtype=torch.float64, device=dev)
    out_i = torch.empty((Q, 10), dtype=torch.int64, device=dev)
    h.score_topk_dev(q.data_ptr(), Q, 10, out_s.data_ptr(), out_i.data_ptr())
    torch.cuda.synchronize()
    wsc, wids = O.topk(O.scores_f64(q.cpu().numpy(), t.cpu().numpy().astype(np.float64)), 10)
    assert np.array_equal(out_i.cpu().numpy(), wids)
    assert np.abs(out_s.cpu().numpy() - wsc).max() < 1e-12
    assert out_i[0, 0].item() == 12345 % N and out_i[0, 1].item() == N - 1


@pytest.mark.parametrize("Q,N,S,k", [(3, 571, 64, 571), (5, 3000, 32, 100), (2, 40, 16, 33)])
def test_large_k_exact_paging(Q, N, S, k):
    """nbest is user-chosen in sse_demo.py:146; any k <= N must work (ties: lower row first)."""
    rng = np.random.RandomState(k)
    q, t = _unit(rng, Q, S), _unit(rng, N, S)
    t[N // 2] = t[1]
    t[N - 1] = t[1]
    h = _scorer()
    h.index_upload(t)
    sc, ids = h.score_topk(q, k)
    wsc, wids = O.topk(O.scores_f64(q, t.astype(np.float64)), k)
    assert np.array_equal(ids, wids)
    assert np.abs(sc - wsc).max() < 1e-12


# --------------------------------------------------------------------------
# option "score_bf16": the candidate pass of the 128-query-block kernel runs on the bf16 matrix pipe; the float64
# re-scoring (with the bound widened to the bf16 rounding) keeps ids and scores EXACT -- same assertions as above
# --------------------------------------------------------------------------

def _scorer_bf16(S=8):
    h = _scorer(S)
    h.set_option("score_bf16", 1)
    return h


@pytest.mark.parametrize("Q,N,S", [(300, 9000, 64), (129, 8192, 256), (600, 32060, 64), (200, 10000, 50), (4000, 70000, 256),
                                   (1, 8200, 32), (1, 200000, 256), (7, 150000, 256), (32, 9000, 64), (3, 300000, 64),
                                   (129, 9000, 128), (5, 20000, 128), (700, 40000, 384)])  # (128: one ring block per tile)
def test_bf16_candidate_pass_keeps_results_exact(Q, N, S):
    rng = np.random.RandomState(Q + N)
    q, t = _unit(rng, Q, S), _unit(rng, N, S)
    h = _scorer_bf16()
    h.index_upload(t)
    k = min(10, N)
    sc, ids = h.score_topk(q, k)
    sub = np.arange(Q) if Q <= 600 else np.random.RandomState(1).choice(Q, 300, replace=False)
    wsc, wids = O.topk(O.scores_f64(q[sub], t.astype(np.float64)), k)
    assert np.array_equal(ids[sub], wids)
    assert np.abs(sc[sub] - wsc).max() < 1e-12
    # and identical to the fp32 candidate pass everywhere
    h.set_option("score_bf16", 0)
    sc32, ids32 = h.score_topk(q, k)
    assert np.array_equal(ids, ids32) and np.array_equal(sc, sc32)


@pytest.mark.parametrize("bf16", [0, 1])
def test_reference_ranking_at_encoding_size_512(bf16):
    """configs[4]: the reference's own ranking (np.dot + getSortedResults, fixture from oracle/make_golden.py) of one
    evaluator batch -- 600 queries x 571 targets x 512 -- reproduced by the HIP scorer: ids exact, float64 scores to 1e-12."""
    from oracle.make_golden import wide_inputs
    z = np.load(os.path.join(G, "scoring_wide512.npz"))
    src, _, tgt64, _ = wide_inputs()
    h = _scorer_bf16() if bf16 else _scorer()
    h.index_upload(tgt64)
    sc, ids = h.score_topk(src, 10)
    assert np.array_equal(ids, z["ranked_idx"][:, :10])
    assert np.abs(sc - z["ranked_score"][:, :10]).max() < 1e-12


def test_bf16_candidate_pass_golden_ties_and_near_ties():
    z = np.load(os.path.join(G, "scoring_eval.npz"))
    h = _scorer_bf16()
    h.index_upload(z["tgt64"])
    sc, ids = h.score_topk(z["src"], 10)
    assert np.array_equal(ids, z["ranked_idx"][:, :10]) and np.abs(sc - z["ranked_score"][:, :10]).max() < 1e-12
    # exact ties + rows that differ below fp32 (let alone bf16) resolution: the certificate fails over to exact paths
    rng = np.random.RandomState(8)
    S, N = 64, 3000
    t = _unit(rng, N, S).astype(np.float64)
    base = t[10].copy()
    for j, r in enumerate((500, 20, 2500, 1234)):
        t[r] = base * (1.0 - (j + 1) * 1e-9)
    t[2999] = t[77]
    q = np.concatenate([base[None, :], t[77:78], _unit(rng, 70, S)]).astype(np.float32)
    h.index_upload(t)
    sc, ids = h.score_topk(q, 5)
    wsc, wids = O.topk(O.scores_f64(q, t), 5)
    assert np.array_equal(ids, wids) and np.abs(sc - wsc).max() < 1e-12


def test_bf16_candidates_second_chance_on_densely_packed_scores():
    """400 index rows whose cosine to a query steps down by 2e-5 -- far closer than the bf16 bound (4e-3), far apart
    for the fp32 one (3e-5): those queries miss the bf16 certificate and take the fp32 second chance (not the float64
    brute force); results stay exact and the ordinary queries are untouched."""
    rng = np.random.RandomState(12)
    S, N, Q = 64, 40000, 300
    t = _unit(rng, N, S).astype(np.float64)
    q = _unit(rng, Q, S)
    for qi in (0, 7, 150):
        rows = rng.choice(N, 400, replace=False)
        base = q[qi].astype(np.float64)
        base /= np.linalg.norm(base)
        for j, r in enumerate(rows):
            u = rng.standard_normal(S)
            u -= u.dot(base) * base
            u /= np.linalg.norm(u)
            c = 1.0 - 2e-5 * j
            t[r] = c * base + np.sqrt(1.0 - c * c) * u
    h = _scorer_bf16()
    h.index_upload(t)
    sc, ids = h.score_topk(q, 10)
    wsc, wids = O.topk(O.scores_f64(q, t), 10)
    assert np.array_equal(ids, wids) and np.abs(sc - wsc).max() < 1e-12
    n = h.get_counter("score_bf16_second_chance_queries")
    assert 3 <= n <= 30, n                                   # the crowded queries took it; the ordinary ones did not


@pytest.mark.parametrize("Q", [1, 5, 32])
def test_few_queries_with_crowded_scores_take_the_collect_pass(Q):
    """<= 32 queries build their fragments inside the sweep (no pack launch) and skip the fp32 second chance: a query whose
    top scores sit closer than the bf16 bound goes straight to the collect pass (one fp32 sweep, final).  Host entry point:
    results come back through the pinned mirror of the re-scoring pass, call after call (completion words carry the call
    number), with certified and uncertified queries in one call."""
    rng = np.random.RandomState(40 + Q)
    S, N = 64, 40000
    t = _unit(rng, N, S).astype(np.float64)
    q = _unit(rng, Q, S)
    rows = rng.choice(N, 400, replace=False)
    base = q[0].astype(np.float64)
    base /= np.linalg.norm(base)
    for j, r in enumerate(rows):
        u = rng.standard_normal(S)
        u -= u.dot(base) * base
        u /= np.linalg.norm(u)
        c = 1.0 - 2e-5 * j
        t[r] = c * base + np.sqrt(1.0 - c * c) * u
    h = _scorer_bf16()
    h.index_upload(t)
    want = O.topk(O.scores_f64(q, t), 10)
    for _ in range(3):
        sc, ids = h.score_topk(q, 10)
        assert np.array_equal(ids, want[1]) and np.abs(sc - want[0]).max() < 1e-12
    assert h.get_counter("score_bf16_second_chance_queries") == 0
    assert 3 <= h.get_counter("score_collect_queries") <= 3 * Q
    assert h.get_counter("score_bruteforce_queries") == 0
    # fewer queries in the next call: words of the earlier, larger call must not be taken for this one's
    sc, ids = h.score_topk(q[Q - 1:], 10)
    assert np.array_equal(ids, want[1][Q - 1:]) and np.abs(sc - want[0][Q - 1:]).max() < 1e-12
    # and the device entry point (every stage queued, no host check in between)
    import torch
    dev = torch.device("cuda:0")
    qd = torch.from_numpy(q).to(dev)
    out_s = torch.empty((Q, 10), dtype=torch.float64, device=dev)
    out_i = torch.empty((Q, 10), dtype=torch.int64, device=dev)
    h.score_topk_dev(qd.data_ptr(), Q, 10, out_s.data_ptr(), out_i.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(out_i.cpu().numpy(), want[1]) and np.abs(out_s.cpu().numpy() - want[0]).max() < 1e-12


@pytest.mark.parametrize("Q,N,S", [(1024, 571, 256), (1100, 1, 256), (2000, 15, 250), (1500, 16, 256), (1030, 17, 249), (1300, 1024, 256),
                                   (1056, 993, 253), (4000, 640, 256), (1025, 641, 256), (1100, 571, 64), (3000, 1000, 57),
                                   (1300, 300, 50), (1024, 33, 49), (1500, 571, 56), (1200, 571, 48), (1200, 571, 65)])
def test_small_index_path_equals_the_list_sweep(Q, N, S):
    """Many queries (>= 1024) against <= 1024 rows at the default encoding size are scored by one launch that forms all N
    scores per query and selects the 16 best by a threshold search (option score_small_index, default on): ids and float64
    scores equal the list sweep's (option off) and the oracle's -- with ties, more exact copies of a query's best row than
    the 16 candidates hold (collect pass), fewer rows than candidates, partial query tiles, un-normalised queries."""
    rng = np.random.RandomState(Q * 7 + N + S)
    q, t = _unit(rng, Q, S), _unit(rng, N, S)
    if N > 40:
        t[N - 1] = t[3]                                      # exact ties: the lower row first
        t[N // 2] = t[3]
        q[0] = t[3]
        for r in rng.choice(N, 30, replace=False):           # 30 exact copies of query 1's best row: more than 16 candidates tie
            t[r] = t[7]
        q[1] = t[7]
    if N > 200:
        for r in rng.choice(N, 100, replace=False):          # and 100 copies: more winners than the selection's 63 lanes
            t[r] = t[11]
        q[2] = t[11]
    q[Q - 1] *= 23.0                                         # sse_demo.py:123 scores with the un-normalised encoding
    h = _scorer()
    h.index_upload(t)
    k = min(10, N)
    sc, ids = h.score_topk(q, k)
    h.set_option("score_small_index", 0)
    sc0, ids0 = h.score_topk(q, k)
    assert np.array_equal(ids, ids0) and np.array_equal(sc, sc0)
    sub = np.concatenate([np.arange(40), np.arange(Q - 40, Q)])
    wsc, wids = O.topk(O.scores_f64(q[sub], t.astype(np.float64)), k)
    assert np.array_equal(ids[sub], wids)
    assert np.abs(sc[sub] - wsc).max() < 1e-9


# --------------------------------------------------------------------------
# index dimensions beyond one 128-query LDS block: the sweep runs with 64-query blocks (296 < S <= 616; BASELINE
# configs[4] has S = 512) or 32-query blocks (S <= 1024).  The reference scores any S with one np.dot
# (sse_evaluator.py:110-111, sse_model.py:286); same exactness bar as everywhere above.
# --------------------------------------------------------------------------

@pytest.mark.parametrize("Q,N,S,bf", [(600, 571, 512, 0), (600, 571, 512, 1), (300, 9000, 297, 0), (300, 9000, 320, 1),
                                      (257, 12000, 616, 1), (130, 8500, 617, 1), (70, 9000, 1024, 1), (1, 9000, 1024, 1),
                                      (33, 40, 700, 0), (7, 150000, 512, 1)])
def test_wide_index_dimensions_exact(Q, N, S, bf):
    rng = np.random.RandomState(Q + N + S)
    q, t = _unit(rng, Q, S), _unit(rng, N, S)
    t[N - 1] = t[3]                                          # an exact tie, the later copy in the last (partial) tile
    q[0] = t[3]
    h = _scorer()
    h.set_option("score_bf16", bf)
    h.index_upload(t)
    k = min(10, N)
    sc, ids = h.score_topk(q, k)
    wsc, wids = O.topk(O.scores_f64(q, t.astype(np.float64)), k)
    assert ids[0, :2].tolist() == [3, N - 1]
    assert np.array_equal(ids, wids)
    assert np.abs(sc - wsc).max() < 1e-12


@pytest.mark.parametrize("k", [10, 40])
def test_configs4_dimension_100k_rows_bf16_equals_fp32_candidates(k):
    """100,000 x 512 index, 2,000 queries (64-query blocks): bf16 and fp32 candidate passes return bit-identical
    ids / scores; a sample is checked against the float64 oracle; k = 40 takes the collect path at S = 512."""
    import torch
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(4)
    N, S, Q = 100_000, 512, 2000
    t = torch.nn.functional.normalize(torch.randn((N, S), generator=g, device=dev), dim=1)
    q = torch.nn.functional.normalize(torch.randn((Q, S), generator=g, device=dev), dim=1)
    rows = torch.randperm(N, generator=g, device=dev)[:Q]
    t[rows] = torch.nn.functional.normalize(q + 0.05 * torch.randn((Q, S), generator=g, device=dev), dim=1)
    h = _scorer()
    h.index_set_dev(t.data_ptr(), N, S)
    res = {}
    for bf in (1, 0):
        h.set_option("score_bf16", bf)
        out_s = torch.empty((Q, k), dtype=torch.float64, device=dev)
        out_i = torch.empty((Q, k), dtype=torch.int64, device=dev)
        h.score_topk_dev(q.data_ptr(), Q, k, out_s.data_ptr(), out_i.data_ptr())
        torch.cuda.synchronize()
        res[bf] = (out_s.cpu().numpy(), out_i.cpu().numpy())
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][0], res[1][0])
    s, i = res[1]
    assert np.array_equal(i[:, 0], rows.cpu().numpy())
    sample = np.random.RandomState(0).choice(Q, 100, replace=False)
    wsc, wids = O.topk(O.scores_f64(q.cpu().numpy()[sample], t.cpu().numpy().astype(np.float64)), k)
    assert np.array_equal(i[sample], wids)
    assert np.abs(s[sample] - wsc).max() < 1e-12
    assert h.get_counter("score_bruteforce_queries") == 0


def test_index_dimension_limit_is_loud():
    import sse_amd
    h = _scorer()
    with pytest.raises(sse_amd.SSEError, match="index dimension"):
        h.index_upload(np.zeros((4, 1025), np.float32))


def test_alternating_shapes_share_the_pinned_mirror_block():
    """The few-queries path stores scores / ids / certificates / completion words through one pinned block whose layout
    moves with (Q, k), and sse_encode's small read-back uses the same block (a stale word of an earlier,
    differently shaped call must never be taken for this call's completion word).  One handle, calls of changing shape
    interleaved with small encodes and a host-buffer encode + score: every result against the float64 oracle."""
    params = model_params("dual-encoder", 300, 20, 64, 64, 64, 9)
    m, p = make_pair(params, seed=5)
    rng = np.random.RandomState(8)
    t = _unit(rng, 9000, 64)
    m.handle.index_upload(t)
    t64 = t.astype(np.float64)
    shapes = [(1, 10), (64, 16), (3, 5), (1, 1), (7, 16), (64, 1), (2, 10), (33, 7), (1, 16), (64, 16), (5, 3)]
    for rep in range(3):
        for Q, k in shapes:
            q = _unit(rng, Q, 64)
            sc, ids = m.handle.score_topk(q, k)
            wsc, wids = O.topk(O.scores_f64(q, t64), k)
            assert np.array_equal(ids, wids), (rep, Q, k)
            assert np.abs(sc - wsc).max() < 1e-12
            n = int(rng.randint(1, 40))
            tok = random_ids(rng, n, 9, 300)
            enc = m.encode_source(tok)                                  # <= 64 KiB: the pinned read-back of sse_encode
            assert np.abs(enc - O.encode(p, params, "src", tok)).max() < 1e-4
            one = random_ids(rng, 1 + rep, 9, 300)
            es, ei = m.handle.encode_score_topk(0, one, True, k)
            w = O.encode(p, params, "src", one)
            _, wi = O.topk(O.scores_f64(w, t64), k)
            assert ei.shape == wi.shape
            margin_ok = True                                             # ids equal unless the oracle's own margin is within the encoder tolerance
            ws, _ = O.topk(O.scores_f64(w, t64), min(k + 1, 9000))
            if ws.shape[1] > 1:
                margin_ok = bool(np.all(np.diff(-ws, axis=1) > 1e-5))
            if margin_ok:
                assert np.array_equal(ei, wi), (rep, Q, k)


@pytest.mark.parametrize("Q,N,S,k", [(4100, 32060, 256, 10), (1100, 9000, 64, 16), (3000, 20011, 128, 1), (2100, 50000, 50, 10)])
def test_two_pass_path_for_mid_size_indexes_is_exact(Q, N, S, k):
    """Indexes of 10^4 .. 10^5 rows under >= 1024 queries (the reference's real evaluation shape: 16,491 x 32,060) are ranked by
    a max-only bf16 sweep -> per-query threshold from the lane maxima -> bf16 collect sweep -> float64 select (option
    score_two_pass_rows) instead of the list sweep.  Same exact ids (ties: lower row first) and float64 scores as the list sweep
    and the oracle: random unit vectors; an exact tie across the index; a query whose best rows are CROWDED inside the bf16 bound
    (hundreds of rows within 1e-4 of each other: the collect buffer overflows and the float64 brute force serves it); an
    id_base; a ragged last tile."""
    rng = np.random.RandomState(Q + N)
    t = _unit(rng, N, S).astype(np.float64)
    q = _unit(rng, Q, S)
    # exact duplicates: tie -> lower row first.  Entries that are multiples of 1/4: every partial sum of the dot is exact in
    # float64, so the three scores are identical in ANY summation order (with generic rows numpy's BLAS gives the duplicates
    # scores that differ in the last bit, by position in the matrix -- the device, one fixed order per row, does not)
    t[17] = rng.choice([-0.25, 0.0, 0.25], size=S)
    t[N - 1] = t[17]
    t[N // 2] = t[17]
    q[3] = t[17].astype(np.float32)
    base = q[5].astype(np.float64)
    base /= np.linalg.norm(base)
    crowd = rng.choice(np.setdiff1d(np.arange(100, N - 100), [N // 2]), 700, replace=False)
    for j, r in enumerate(crowd):                              # 700 rows within 7e-5 of the top score of query 5
        u = rng.standard_normal(S)
        u -= u.dot(base) * base
        u /= np.linalg.norm(u)
        c = 1.0 - 1e-7 * j
        t[r] = c * base + np.sqrt(max(0.0, 1.0 - c * c)) * u
    h = _scorer_bf16()
    h.index_upload(t, id_base=1000)
    wsc, wids = O.topk(O.scores_f64(q, t), k)
    h.set_option("score_two_pass_rows", 0)
    sc0, ids0 = h.score_topk(q, k)
    assert np.array_equal(ids0, wids + 1000) and np.abs(sc0 - wsc).max() < 1e-12
    h.set_option("score_two_pass_rows", 524288)
    h.set_option("score_two_pass_min_rows", 0)
    b0, n0 = h.get_counter("score_bruteforce_queries"), h.get_counter("score_two_pass_calls")
    sc1, ids1 = h.score_topk(q, k)
    assert h.get_counter("score_two_pass_calls") == n0 + 1    # the path under test did run
    assert np.array_equal(ids1, wids + 1000)
    assert np.array_equal(sc1, sc0)                           # float64 scores of the same rows in the same arithmetic: identical bits
    assert list(ids1[3, :min(k, 3)]) == [1017, 1000 + N // 2, 1000 + N - 1][:k]
    assert 1 <= h.get_counter("score_bruteforce_queries") - b0 <= 12  # the crowded query (+ the few whose top scores sit within the bf16 bound by chance)
    # device-buffer entry point, asynchronous on a side stream
    import torch
    dev = torch.device("cuda", 0)
    qd = torch.from_numpy(q).to(dev)
    os_ = torch.empty((Q, k), dtype=torch.float64, device=dev)
    oi_ = torch.empty((Q, k), dtype=torch.int64, device=dev)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        h.score_topk_dev(qd.data_ptr(), Q, k, os_.data_ptr(), oi_.data_ptr(), st.cuda_stream)
    st.synchronize()
    assert np.array_equal(oi_.cpu().numpy(), wids + 1000) and np.array_equal(os_.cpu().numpy(), sc0)
    assert h.get_counter("score_two_pass_calls") == n0 + 2

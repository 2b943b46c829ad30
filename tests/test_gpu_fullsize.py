"""Full-size configurations of BASELINE.json / SURVEY 8d, HIP (through the C ABI) against the CPU oracle.

The kernels change launch shape with size (index split policy, 64- vs 32-row LSTM tiles, dK slices), so small-shape
parity does not cover them.  Each test names the config it is the full size of.  Where the float64 oracle over the
WHOLE problem would take minutes (C4: 1,000 x 1 M x 256), every query is checked against an independent float64
product on the device (torch) and a sample against the numpy oracle; the sizes are stated in the test."""
import os
import time

import numpy as np
import pytest

from oracle import sse_oracle as O
from tests.util import make_pair, model_params, random_ids

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _oracle_topk_batched(src, tgt64, k, batch=600):
    """Evaluator-style batches (sse_evaluator.py:104-112) through the oracle's scorer."""
    ids, sc = [], []
    for b0 in range(0, len(src), batch):
        s, i = O.topk_fast(O.scores_f64(src[b0:b0 + batch], tgt64), k)     # == O.topk (tests/test_oracle.py)
        sc.append(s)
        ids.append(i)
    return np.concatenate(sc), np.concatenate(ids)


def test_c3_crosslingual_full_index_and_queries():
    """configs[2] / SURVEY 8d C3 as specified: ALL 32,060 targets indexed and ALL 16,491 queries scored, on the
    token-id rows the reference's prepare_raw_data produced (tests/golden/crosslingual_full_ids.npz), dual-encoder
    H = S = 256, E = 50, T = 50 (makefile:42 + BASELINE).  Encodings within 1e-3 (north_star; observed ~1e-6), top-10
    ids of every query and the top-1/3/10 accuracies exact against the oracle."""
    z = np.load(os.path.join(G, "crosslingual_full_ids.npz"))
    src_ids, tgt_ids = z["src_ids"].astype(np.int32), z["tgt_ids"].astype(np.int32)
    labels = [[int(v) for v in row if v >= 0] for row in z["labels"]]
    V = int(z["vocab_size"])
    params = model_params("dual-encoder", V, 50, 256, 256, 256, 50)
    m, p = make_pair(params, seed=21)
    tgt = m.encode_target(tgt_ids)
    src = m.encode_source(src_ids)
    want_tgt = O.encode(p, params, "tgt", tgt_ids)
    want_src = O.encode(p, params, "src", src_ids)
    assert tgt.shape == (32060, 256) and src.shape == (16491, 256)
    assert np.abs(tgt - want_tgt).max() < 1e-3 and np.abs(src - want_src).max() < 1e-3
    assert np.abs(tgt - want_tgt).max() < 2e-5 and np.abs(src - want_src).max() < 2e-5      # what fp32 actually gives
    # scoring on IDENTICAL encodings (the oracle's), so that the id comparison is bit-exact by construction
    t64 = want_tgt.astype(np.float64)
    m.handle.index_upload(t64)
    sc, ids = m.handle.score_topk(want_src, 10)
    wsc, wids = _oracle_topk_batched(want_src, t64, 10)
    assert np.array_equal(ids, wids)
    assert np.abs(sc - wsc).max() < 1e-12
    for n in (1, 3, 10):
        assert O.topk_tight_accuracy(n, labels, ids) == O.topk_tight_accuracy(n, labels, wids)
    # and end to end on the device's own encodings: cosine within 1e-3, top-1 id equal wherever the oracle's top-2
    # margin exceeds the encoding tolerance
    m.handle.index_upload(tgt.astype(np.float64))
    sc2, ids2 = m.handle.score_topk(src, 10)
    assert np.abs(sc2[:, 0] - wsc[:, 0]).max() < 1e-3
    clear = (wsc[:, 0] - wsc[:, 1]) > 1e-4
    assert np.array_equal(ids2[clear, 0], wids[clear, 0])
    # the observed agreement, all queries (shown with pytest -s / in the captured output of a failure)
    print("C3 crosslingual: top-1 id agreement device encodings vs oracle encodings %.5f (%d of %d queries; %d with a top-2 "
          "margin above 1e-4, all of those equal); top-10 set agreement %.5f; max |cosine diff| %.2e"
          % (np.mean(ids2[:, 0] == wids[:, 0]), int(np.sum(ids2[:, 0] == wids[:, 0])), len(wids), int(clear.sum()),
             np.mean([len(set(a) & set(b)) / 10.0 for a, b in zip(ids2, wids)]), np.abs(sc2[:, 0] - wsc[:, 0]).max()))
    # north_star: "top-1 id bit-exact".  Observed on this data: 16491 of 16491 (profiles/r03x_c3_agreement.txt).  A query
    # may legitimately flip only where the oracle's own top-2 margin is below the encoder tolerance (observed max cosine
    # difference 1.8e-7; the 94 % of the queries whose margin exceeds 1e-4 are asserted equal above, and so are all those
    # above 10x the observed difference); pin the overall count so that a regression shows.
    assert clear.mean() > 0.9
    clear5 = (wsc[:, 0] - wsc[:, 1]) > 2e-6
    assert np.array_equal(ids2[clear5, 0], wids[clear5, 0])
    assert int(np.sum(ids2[:, 0] == wids[:, 0])) >= len(wids) - 1          # agreement >= 0.9999
    assert np.mean([len(set(a) & set(b)) / 10.0 for a, b in zip(ids2, wids)]) > 0.9999


def test_qna_real_data_T1000():
    """rawdata-qna as the reference prepares it (makefile:17: max_seq_length = 1000, vocab 8000 -> 2737 subwords):
    93 targets and 602 queries of up to 999 tokens, reference default cell sizes (96 / 96, E = 50, S = 64)."""
    z = np.load(os.path.join(G, "qna_full_ids.npz"))
    src_ids, tgt_ids = z["src_ids"].astype(np.int32), z["tgt_ids"].astype(np.int32)
    assert src_ids.shape == (602, 1000) and tgt_ids.shape == (93, 1000)
    params = model_params("dual-encoder", int(z["vocab_size"]), 50, 96, 96, 64, 1000)
    m, p = make_pair(params, seed=4)
    tgt, src = m.encode_target(tgt_ids), m.encode_source(src_ids)
    want_tgt, want_src = O.encode(p, params, "tgt", tgt_ids), O.encode(p, params, "src", src_ids)
    assert np.abs(tgt - want_tgt).max() < 1e-4 and np.abs(src - want_src).max() < 1e-4
    t64 = want_tgt.astype(np.float64)
    m.handle.index_upload(t64)
    sc, ids = m.handle.score_topk(want_src, 10)
    wsc, wids = O.topk(O.scores_f64(want_src, t64), 10)
    assert np.array_equal(ids, wids) and np.abs(sc - wsc).max() < 1e-12
    labels = [[int(v) for v in row if v >= 0] for row in z["labels"]]
    assert O.topk_tight_accuracy(1, labels, ids) == O.topk_tight_accuracy(1, labels, wids)


def _c4_slice(N, Q, S=256, seed=1):
    """unit-normal rows, a planted neighbour normalize(q + 0.1 noise) per query at a known row."""
    import torch
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(seed)
    t = torch.nn.functional.normalize(torch.randn((N, S), generator=g, device=dev), dim=1)
    q = torch.nn.functional.normalize(torch.randn((Q, S), generator=g, device=dev), dim=1)
    rows = torch.randperm(N, generator=g, device=dev)[:Q]
    t[rows] = torch.nn.functional.normalize(q + 0.1 * torch.randn((Q, S), generator=g, device=dev), dim=1)
    return t, q, rows


@pytest.mark.parametrize("bf16", [0, 1])
def test_c4_slice_1M_rows_1000_queries(bf16):
    """configs[3] correctness slice ('1M x 1k slice vs oracle f64'), with fp32 and with bf16 candidates.
    All 1,000 queries: ids equal to a float64 product + top-k computed independently on the device (torch), scores
    within 1e-12; 24 of them additionally against the numpy oracle (the reference's arithmetic), exactly."""
    import torch
    from tests.test_gpu_score import _scorer
    N, Q, S, k = 1_000_000, 1000, 256, 10
    t, q, rows = _c4_slice(N, Q)
    h = _scorer()
    h.set_option("score_bf16", bf16)
    h.index_set_dev(t.data_ptr(), N, S)
    out_s = torch.empty((Q, k), dtype=torch.float64, device=t.device)
    out_i = torch.empty((Q, k), dtype=torch.int64, device=t.device)
    h.score_topk_dev(q.data_ptr(), Q, k, out_s.data_ptr(), out_i.data_ptr())
    torch.cuda.synchronize()
    s, i = out_s.cpu().numpy(), out_i.cpu().numpy()
    assert np.array_equal(i[:, 0], rows.cpu().numpy())                       # planted neighbours found
    assert np.all(np.diff(s, axis=1) <= 0) and all(len(set(r)) == k for r in i)
    # independent float64 reference on the device, in query chunks (250 x 1M x 8 B = 2 GB each)
    t64 = t.double()
    for c0 in range(0, Q, 250):
        ref = q[c0:c0 + 250].double() @ t64.T
        rs, ri = torch.topk(ref, k, dim=1, largest=True, sorted=True)
        rs, ri = rs.cpu().numpy(), ri.cpu().numpy()
        assert np.abs(s[c0:c0 + 250] - rs).max() < 1e-12
        same = i[c0:c0 + 250] == ri
        # a different summation order may swap neighbours closer than 1e-13; nothing else may differ
        gap_ok = np.abs(s[c0:c0 + 250] - rs) < 1e-13
        assert np.all(same | gap_ok) and same.mean() > 0.999
        del ref
    sample = np.random.RandomState(0).choice(Q, 24, replace=False)
    wsc, wids = O.topk_fast(O.scores_f64(q.cpu().numpy()[sample], t.cpu().numpy().astype(np.float64)), k)
    assert np.array_equal(i[sample], wids) and np.abs(s[sample] - wsc).max() < 1e-12
    assert h.get_counter("score_bruteforce_queries") == 0


def _timed(fn, reps=3):
    import torch
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def test_duplicates_of_the_best_row_stay_fast_and_exact():
    """A 1 M-row index holding 64 exact duplicates of each of 100 queries' best rows (real catalogues have them): the
    k-th score is tied with rows outside every candidate list, so no certificate can hold.  The collect path serves
    those queries with one more grid-wide sweep -- exact ids (ties: lower row first), no float64 brute force, and the
    call stays within 10x of the same call on a duplicate-free index (round 1: one workgroup per query sweeping all
    N rows in float64, seconds)."""
    import torch
    from tests.test_gpu_score import _scorer
    N, Q, S, k = 1_000_000, 100, 256, 10
    t, q, rows = _c4_slice(N, Q, seed=3)
    h = _scorer()
    h.index_set_dev(t.data_ptr(), N, S)
    out_s = torch.empty((Q, k), dtype=torch.float64, device=t.device)
    out_i = torch.empty((Q, k), dtype=torch.int64, device=t.device)
    base = _timed(lambda: h.score_topk_dev(q.data_ptr(), Q, k, out_s.data_ptr(), out_i.data_ptr()))
    # the 64 copies of a query sit NEXT TO each other (variants of one item are adjacent in a catalogue): one lane list
    # of the sweep sees 16 of them, more than it keeps, so the bound equals the k-th score and no certificate can hold
    g = torch.Generator(device=t.device).manual_seed(9)
    starts = torch.randperm(N // 64, generator=g, device=t.device)[:Q] * 64
    dup = starts[:, None] + torch.arange(64, device=t.device)[None, :]
    t[dup.reshape(-1)] = q.repeat_interleave(64, dim=0)               # 64 rows == the query itself (cosine 1.0)
    h.index_set_dev(t.data_ptr(), N, S)
    c0, b0 = h.get_counter("score_collect_queries"), h.get_counter("score_bruteforce_queries")
    dt = _timed(lambda: h.score_topk_dev(q.data_ptr(), Q, k, out_s.data_ptr(), out_i.data_ptr()), reps=2)
    i = out_i.cpu().numpy()
    want = np.sort(dup.cpu().numpy(), axis=1)[:, :k]                  # the 10 lowest of the 64 tied rows
    assert np.array_equal(i, want)
    s = out_s.cpu().numpy()
    qn, tn = q.double(), t[dup[:, 0]].double()
    assert np.abs(s - (qn * tn).sum(dim=1, keepdim=True).cpu().numpy()).max() < 1e-12
    assert h.get_counter("score_collect_queries") - c0 >= Q            # served by the collect path ...
    assert h.get_counter("score_bruteforce_queries") == b0             # ... not by the brute force
    assert dt < 10 * base + 2e-3, (dt, base)


@pytest.mark.parametrize("Q", [1, 200])
def test_k100_on_1M_rows(Q):
    """nbest is user-chosen (sse_demo.py:128-134,146; webserver.py): k = 100 over 1 M rows through the collect path,
    exact against the oracle for up to 16 queries and against a device float64 top-k for all, within 10x of k = 10."""
    import torch
    from tests.test_gpu_score import _scorer
    N, S, k = 1_000_000, 256, 100
    t, q, rows = _c4_slice(N, max(Q, 1), seed=5)
    h = _scorer()
    h.index_set_dev(t.data_ptr(), N, S)
    o10s = torch.empty((Q, 10), dtype=torch.float64, device=t.device)
    o10i = torch.empty((Q, 10), dtype=torch.int64, device=t.device)
    out_s = torch.empty((Q, k), dtype=torch.float64, device=t.device)
    out_i = torch.empty((Q, k), dtype=torch.int64, device=t.device)
    base = _timed(lambda: h.score_topk_dev(q.data_ptr(), Q, 10, o10s.data_ptr(), o10i.data_ptr()))
    b0 = h.get_counter("score_bruteforce_queries")
    dt = _timed(lambda: h.score_topk_dev(q.data_ptr(), Q, k, out_s.data_ptr(), out_i.data_ptr()))
    s, i = out_s.cpu().numpy(), out_i.cpu().numpy()
    assert np.array_equal(i[:, :10], o10i.cpu().numpy()) and np.array_equal(s[:, :10], o10s.cpu().numpy())
    ref = q.double() @ t.double().T
    rs, ri = torch.topk(ref, k, dim=1)
    assert np.abs(s - rs.cpu().numpy()).max() < 1e-12
    same = i == ri.cpu().numpy()
    assert np.all(same | (np.abs(s - rs.cpu().numpy()) < 1e-13))
    n = min(Q, 16)
    wsc, wids = O.topk_fast(O.scores_f64(q.cpu().numpy()[:n], t.cpu().numpy().astype(np.float64)), k)
    assert np.array_equal(i[:n], wids) and np.abs(s[:n] - wsc).max() < 1e-12
    assert h.get_counter("score_bruteforce_queries") == b0
    assert dt < 10 * base + 2e-3, (dt, base)


def test_c2_lstm_train_step_8192_rows():
    """configs[1] training at the large batch bench.py times (8192 pair rows, dual-encoder H = S = 256, T = 32):
    64-row forward tiles, 256 BPTT tiles, 64 dK slices.  Same tolerances as the small-shape step test."""
    V, E, H, S, T, B = 2000, 50, 256, 256, 32, 8192
    params = model_params("dual-encoder", V, E, H, H, S, T, lr=0.9)
    m, p = make_pair(params, seed=8)
    st = O.new_optimizer_state(p)
    rng = np.random.RandomState(13)
    src = np.repeat(random_ids(rng, B // 2, T, V, 0.5), 2, axis=0)
    tgt = random_ids(rng, B, T, V, 0.5)
    z = np.tile(np.array([1.0, 0.0], np.float32), B // 2)
    want_loss, want_acc = O.train_step(p, st, params, src, tgt, z, 0.9)
    loss, acc = m.train_step(src, tgt, z)
    assert loss == pytest.approx(float(want_loss), rel=1e-5, abs=1e-6)
    assert acc == pytest.approx(float(want_acc), abs=1e-6)
    got = m.get_variables(with_slots=True)
    for name, w in p.items():
        assert np.abs(got[name].reshape(w.shape) - w).max() < 2e-4, name
        assert np.abs(got[name + "/Adagrad"].reshape(w.shape) - st[name]).max() < 2e-4, name + "/Adagrad"


def test_c5_cnn_train_step_1024_rows():
    """configs[4] at its global batch of 1024 pair rows (T = 64, S = 512, E = 50, 571 target rows; fp32 arithmetic).
    The loss is builder-defined (the reference's CNN graph does not build): oracle._cnn_gradients."""
    V, E, S, T, B, N = 3000, 50, 512, 64, 1024, 571
    params = model_params("source_only_cnn", V, E, 96, 96, S, T, N=N, lr=0.9)
    m, p = make_pair(params, seed=6)
    st = O.new_optimizer_state(p)
    rng = np.random.RandomState(3)
    src = np.repeat(random_ids(rng, B // 2, T, V, 0.5), 2, axis=0)
    rows = rng.randint(0, N, size=B).astype(np.int32)
    z = np.tile(np.array([1.0, 0.0], np.float32), B // 2)
    want_loss, want_acc = O.train_step(p, st, params, src, rows, z, 0.9)
    loss, acc = m.train_step(src, rows, z)
    assert loss == pytest.approx(float(want_loss), rel=1e-5, abs=1e-6)
    assert acc == pytest.approx(float(want_acc), abs=1e-6)
    got = m.get_variables(with_slots=True)
    for name, w in p.items():
        assert np.abs(got[name].reshape(w.shape) - w).max() < 5e-4, name
        assert np.abs(got[name + "/Adagrad"].reshape(w.shape) - st[name]).max() < 5e-4, name + "/Adagrad"


def test_c5_cnn_train_step_8192_rows():
    """configs[4] at its LARGE global batch (B_rows in {1024, 8192}): 8192 pair rows, T = 64, S = 512, E = 50,
    571 target rows, fp32.  2048 chunks of the gather / scatter backward, 32 projection-backward chunks; same tolerances as
    the 1024-row step.  (The oracle's step takes ~40 s of host time here.)"""
    V, E, S, T, B, N = 3000, 50, 512, 64, 8192, 571
    params = model_params("source_only_cnn", V, E, 96, 96, S, T, N=N, lr=0.9)
    m, p = make_pair(params, seed=16)
    st = O.new_optimizer_state(p)
    rng = np.random.RandomState(23)
    src = np.repeat(random_ids(rng, B // 2, T, V, 0.5), 2, axis=0)
    rows = rng.randint(0, N, size=B).astype(np.int32)
    z = np.tile(np.array([1.0, 0.0], np.float32), B // 2)
    want_loss, want_acc = O.train_step(p, st, params, src, rows, z, 0.9)
    loss, acc = m.train_step(src, rows, z)
    assert loss == pytest.approx(float(want_loss), rel=1e-5, abs=1e-6)
    assert acc == pytest.approx(float(want_acc), abs=1e-6)
    got = m.get_variables(with_slots=True)
    for name, w in p.items():
        assert np.abs(got[name].reshape(w.shape) - w).max() < 5e-4, name
        assert np.abs(got[name + "/Adagrad"].reshape(w.shape) - st[name]).max() < 5e-4, name + "/Adagrad"


def test_c2_encode_131072_rows():
    """configs[1] at the largest batch of SURVEY 8d C2's sweep (B = 131072; 2048 workgroups of 64 rows = 8 rounds over the
    256 CUs), dense ids, device-resident in and out (sse_encode_dev).  Size-independent property: a row's encoding does not
    depend on the batch it sits in -- the 131072 rows are 32 shuffled copies of 4096 distinct sequences; every copy must
    be BIT-identical to the first, and the 4096 distinct ones are checked against the oracle."""
    import torch
    V, E, H, S, T, B, D = 32000, 50, 256, 256, 32, 131072, 4096
    params = model_params("dual-encoder", V, E, H, H, S, T)
    m, p = make_pair(params, seed=2)
    rng = np.random.RandomState(77)
    base = random_ids(rng, D, T, V)                                  # dense (no PAD): the bench's worst case
    base[::7, :5] = 0                                                 # ... plus some left-padded rows
    perm = np.concatenate([rng.permutation(D) for _ in range(B // D)])
    ids = torch.from_numpy(base[perm]).cuda()
    for side, enc in ((0, "src"), (1, "tgt")):
        out = torch.empty((B, S), dtype=torch.float32, device="cuda")
        m.handle.encode_dev(side, ids.data_ptr(), B, T, True, out.data_ptr())
        m.handle.synchronize()
        got = out.cpu().numpy()
        first = np.empty((D, S), np.float32)
        first[perm[:D]] = got[:D]
        assert np.array_equal(got, first[perm]), enc                  # batch position never changes a bit
        want = O.encode(p, params, enc, base)
        assert np.abs(first - want).max() <= 1e-4
        assert np.sum(first.astype(np.float64) * want, axis=1).min() > 1 - 1e-6

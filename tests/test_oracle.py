"""Pins for the CPU oracle (the model part is 'parity unpinned' vs TF; see
oracle/sse_oracle.py header): torch.nn.LSTM cross-check, finite differences."""
import numpy as np
import pytest
import torch

from oracle import sse_oracle as O


def _cfg(mode="dual-encoder", V=97, E=10, H=12, S=8, Ht=None):
    return dict(vocab_size=V, embedding_size=E, encoding_size=S, src_cell_size=H,
                tgt_cell_size=Ht or H, network_mode=mode, targetSpaceSize=5)


def _torch_lstm_last(emb, kernel, bias, ids):
    """torch.nn.LSTM with TF weights remapped: TF kernel cols [i|j|f|o] ->
    torch rows [i|f|g|o]; forget_bias 1.0 folded into bias_ih's f block."""
    E = emb.shape[1]
    H = kernel.shape[1] // 4
    lstm = torch.nn.LSTM(E, H, batch_first=True)
    order = [0, 2, 1, 3]
    blocks = [kernel[:, k * H:(k + 1) * H] for k in range(4)]
    W = np.concatenate([blocks[k] for k in order], axis=1)
    bb = [bias[k * H:(k + 1) * H].copy() for k in range(4)]
    bb[2] = bb[2] + 1.0
    b = np.concatenate([bb[k] for k in order])
    with torch.no_grad():
        lstm.weight_ih_l0.copy_(torch.from_numpy(W[:E].T.copy()))
        lstm.weight_hh_l0.copy_(torch.from_numpy(W[E:].T.copy()))
        lstm.bias_ih_l0.copy_(torch.from_numpy(b))
        lstm.bias_hh_l0.zero_()
        x = torch.from_numpy(emb[ids])
        out, _ = lstm(x)
    return out[:, -1].numpy()


@pytest.mark.parametrize("E,H,T,B", [(10, 12, 7, 5), (50, 96, 20, 9), (50, 256, 32, 4)])
def test_lstm_matches_torch(E, H, T, B):
    cfg = _cfg(E=E, H=H)
    p = O.init_params(cfg, seed=3)
    p["source_encoder/rnn/basic_lstm_cell/bias"] = np.random.RandomState(1).uniform(-.2, .2, 4 * H).astype(np.float32)
    ids = np.random.RandomState(2).randint(0, cfg["vocab_size"], size=(B, T)).astype(np.int32)
    K, b = p["source_encoder/rnn/basic_lstm_cell/kernel"], p["source_encoder/rnn/basic_lstm_cell/bias"]
    h = O.lstm_forward(p["word_embedding"], K, b, ids)
    ht = _torch_lstm_last(p["word_embedding"], K, b, ids)
    assert np.abs(h - ht).max() < 2e-6


def test_l2_normalize_clamp():
    x = np.zeros((2, 4), np.float32)
    x[1] = [3, 0, 4, 0]
    n = O.l2_normalize(x)
    assert np.all(n[0] == 0)
    assert np.allclose(n[1], [0.6, 0, 0.8, 0])
    tiny = np.full((1, 4), 1e-8, np.float32)          # sum sq = 4e-16 < 1e-12 -> scaled by 1e6
    assert np.allclose(O.l2_normalize(tiny), 1e-2, rtol=1e-5)


def test_loss_matches_closed_form():
    rng = np.random.RandomState(0)
    ns, nt = O.l2_normalize(rng.randn(6, 8)), O.l2_normalize(rng.randn(6, 8))
    z = np.array([1, 0, 1, 0, 1, 0], np.float32)
    loss, acc, cos = O.loss_and_acc(ns, nt, z)
    x = 64.0 * np.sum(ns.astype(np.float64) * nt, axis=1)
    ref = np.mean(-z * np.log(1 / (1 + np.exp(-x))) - (1 - z) * np.log(1 - 1 / (1 + np.exp(-x)) + 1e-300))
    assert abs(loss - ref) < 1e-4 * max(1, abs(ref))
    assert 0.0 <= acc <= 1.0


@pytest.mark.parametrize("mode", ["dual-encoder", "shared-encoder"])
def test_gradients_finite_difference(mode):
    cfg = _cfg(mode=mode, V=23, E=5, H=6, S=4)
    p = {k: v.astype(np.float64) for k, v in O.init_params(cfg, seed=1).items()}
    rng = np.random.RandomState(5)
    B, T = 6, 4
    src = rng.randint(0, 23, size=(B, T)).astype(np.int32)
    tgt = rng.randint(0, 23, size=(B, T)).astype(np.int32)
    z = np.array([1, 0] * 3, np.float32)
    # float64 evaluation of the same formulas through the oracle (F32 casts are
    # idempotent on the structure; run the loss in float64 by monkeypatching F32)
    old = O.F32
    try:
        O.F32 = np.float64
        O.FORGET_BIAS, O.L2_EPS, O.LOGIT_SCALE = np.float64(1.0), np.float64(1e-12), np.float64(64.0)

        def f(pp):
            ns = O.encode(pp, cfg, "src", src)
            nt = O.encode(pp, cfg, "tgt", tgt)
            return float(O.loss_and_acc(ns, nt, z.astype(np.float64))[0])

        loss, _, grads = O.gradients(p, cfg, src, tgt, z.astype(np.float64))
        assert abs(loss - f(p)) < 1e-9
        for name, g in grads.items():
            if isinstance(g, tuple):
                g = O.dense_embedding_grad(g, cfg["vocab_size"])
            for _ in range(6):
                idx = tuple(rng.randint(0, s) for s in p[name].shape)
                q = {k: v.copy() for k, v in p.items()}
                eps = 1e-6
                q[name][idx] += eps
                up = f(q)
                q[name][idx] -= 2 * eps
                dn = f(q)
                num = (up - dn) / (2 * eps)
                assert abs(num - g[idx]) < 1e-5 * max(1.0, abs(num)), (name, idx, num, g[idx])
    finally:
        O.F32 = old
        O.FORGET_BIAS, O.L2_EPS, O.LOGIT_SCALE = old(1.0), old(1e-12), old(64.0)


def test_cnn_bf16_gradients_are_the_fp32_gradients_at_the_rounded_weights():
    """cfg["cnn_bf16"] (mixed-precision CNN training, BASELINE configs[4]): forward on bf16-rounded embeddings and
    filters, straight-through backward.  By construction that is the float32 gradient evaluated at the rounded
    weights -- checked bit for bit, together with the update touching the float32 MASTER weights."""
    cfg = _cfg(mode="source_only_cnn", V=40, E=6, H=6, S=8)
    cfg["targetSpaceSize"] = 7
    p = O.init_params(cfg, seed=5)
    rng = np.random.RandomState(1)
    B, T = 8, 9
    src = rng.randint(0, 40, size=(B, T)).astype(np.int32)
    src[0, :5] = 0
    rows = rng.randint(0, 7, size=B).astype(np.int32)
    z = np.array([1, 0] * 4, np.float32)
    rounded = {k: (O.bf16_round(v) if (k == "word_embedding" or k.endswith("/W")) else v.copy()) for k, v in p.items()}
    assert any(not np.array_equal(rounded[k], p[k]) for k in p)
    l16, a16, g16 = O.gradients(p, dict(cfg, cnn_bf16=True), src, rows, z)
    l32, a32, g32 = O.gradients(rounded, cfg, src, rows, z)
    assert l16 == l32 and a16 == a32 and set(g16) == set(g32)
    for name in g16:
        a, b = g16[name], g32[name]
        if isinstance(a, tuple):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), name
        else:
            assert np.array_equal(a, b), name
    # and it is a different function from the float32 one
    assert O.gradients(p, cfg, src, rows, z)[0] != l16
    # the optimizer updates the float32 masters, not the rounded copies
    st = O.new_optimizer_state(p)
    q = {k: v.copy() for k, v in p.items()}
    O.train_step(q, st, dict(cfg, cnn_bf16=True), src, rows, z, 0.5)
    w = "source_only_cnn/conv-maxpool-3/W"
    assert not np.array_equal(q[w], p[w]) and not np.array_equal(O.bf16_round(q[w]), q[w])


def test_cnn_gradients_finite_difference():
    """Builder-defined CNN pair loss (oracle._cnn_gradients): analytic gradients vs central differences, float64.
    All-PAD windows make exact max-pool ties; FD is taken at points where the arg-max is stable."""
    cfg = _cfg(mode="source_only_cnn", V=19, E=4, H=6, S=5)
    cfg["targetSpaceSize"] = 7
    p = {k: v.astype(np.float64) for k, v in O.init_params(cfg, seed=2).items()}
    rng = np.random.RandomState(3)
    B, T = 6, 7
    src = rng.randint(0, 19, size=(B, T)).astype(np.int32)
    src[0, :4] = 0                                                   # left padding: tied windows
    rows = rng.randint(0, 7, size=B).astype(np.int32)
    rows[1] = rows[0]                                                # duplicate target row: slices are summed
    z = np.array([1, 0] * 3, np.float64)
    old = O.F32
    try:
        O.F32 = np.float64
        O.L2_EPS, O.LOGIT_SCALE = np.float64(1e-12), np.float64(64.0)

        def f(pp):
            ns = O.encode(pp, cfg, "src", src)
            nt = O.l2_normalize(pp["target_embedding/tgt_seq_embedding"][rows])
            return float(O.loss_and_acc(ns, nt, z)[0])

        loss, _, grads = O.gradients(p, cfg, src, rows, z)
        assert abs(loss - f(p)) < 1e-9
        assert set(grads) == set(p)
        for name, g in grads.items():
            if isinstance(g, tuple):
                g = O.dense_embedding_grad(g, p[name].shape[0])
            for _ in range(8):
                idx = tuple(rng.randint(0, s) for s in p[name].shape)
                q = {k: v.copy() for k, v in p.items()}
                eps = 1e-7
                q[name][idx] += eps
                up = f(q)
                q[name][idx] -= 2 * eps
                dn = f(q)
                num = (up - dn) / (2 * eps)
                assert abs(num - g[idx]) < 2e-5 * max(1.0, abs(num)), (name, idx, num, g[idx])
    finally:
        O.F32 = old
        O.L2_EPS, O.LOGIT_SCALE = old(1e-12), old(64.0)


def test_cnn_train_step_runs_and_updates_only_touched_rows():
    cfg = _cfg(mode="source_only_cnn", V=40, E=6, H=6, S=8)
    cfg["targetSpaceSize"] = 9
    p = O.init_params(cfg, seed=4)
    before = {k: v.copy() for k, v in p.items()}
    st = O.new_optimizer_state(p)
    rng = np.random.RandomState(0)
    src = rng.randint(2, 30, size=(8, 9)).astype(np.int32)
    rows = np.array([0, 3, 3, 5, 0, 1, 2, 8], np.int32)
    z = np.array([1, 0] * 4, np.float32)
    losses = [float(O.train_step(p, st, cfg, src, rows, z, 0.1)[0]) for _ in range(15)]
    assert min(losses) < losses[0]
    tbl = "target_embedding/tgt_seq_embedding"
    untouched = [r for r in range(9) if r not in set(rows.tolist())]
    assert np.array_equal(p[tbl][untouched], before[tbl][untouched])
    assert np.array_equal(p["word_embedding"][30:], before["word_embedding"][30:])
    assert not np.array_equal(p[tbl][3], before[tbl][3])


def test_train_step_reduces_loss_and_dedups_embedding_rows():
    cfg = _cfg(V=31, E=6, H=8, S=6)
    p = O.init_params(cfg, seed=2)
    st = O.new_optimizer_state(p)
    rng = np.random.RandomState(0)
    src = rng.randint(0, 31, size=(8, 5)).astype(np.int32)
    tgt = rng.randint(0, 31, size=(8, 5)).astype(np.int32)
    src[:, 0] = 0                                        # duplicate PAD ids
    z = np.array([1, 0] * 4, np.float32)
    before = p["word_embedding"].copy()
    losses = [O.train_step(p, st, cfg, src, tgt, z, 0.5)[0] for _ in range(30)]
    assert losses[-1] < losses[0]
    untouched = np.setdiff1d(np.arange(31), np.unique(np.concatenate([src.ravel(), tgt.ravel()])))
    assert np.array_equal(before[untouched], p["word_embedding"][untouched])
    assert np.all(st["word_embedding"][untouched] == O.ADAGRAD_INIT_ACC)


def test_cnn_forward_shape_and_max_semantics():
    cfg = _cfg(mode="source_only_cnn", V=40, E=7, S=9)
    p = O.init_params(cfg, seed=0)
    ids = np.random.RandomState(0).randint(0, 40, size=(3, 11)).astype(np.int32)
    pool = O.cnn_forward(p, ids)
    assert pool.shape == (3, 576) and np.all(pool >= 0)
    # brute force one feature
    fs, nf = 3, 128
    W = p["source_only_cnn/conv-maxpool-3/W"]
    x = p["word_embedding"][ids[1]]
    vals = [max(0.0, float(np.sum(x[s:s + fs] * W[:, :, 0, 5]) + 0.1)) for s in range(11 - fs + 1)]
    assert abs(pool[1, 256 + 5] - max(vals)) < 1e-5
    out = O.encode(p, cfg, "src", ids)
    assert np.allclose(np.linalg.norm(out, axis=1), 1.0, atol=1e-5)


def test_pad_tokens_rules():
    assert O.pad_tokens([5, 6], 6) == [0, 0, 0, 5, 6, 1]
    assert O.pad_tokens([5, 6, 7, 8], 6) == [0, 5, 6, 7, 8, 1]
    assert O.pad_tokens([5, 6, 7, 8, 9], 6) == [0, 5, 6, 7, 8, 1]
    assert O.pad_tokens([], 4) == [0, 0, 0, 1]


def test_index_line_roundtrip_is_float32_exact():
    v = np.random.RandomState(0).randn(16).astype(np.float32)
    line = O.format_index_line("id7", "Some Title", v)
    ids, sents, enc = O.parse_index_lines([line, "bad line\n"])
    assert ids == ["id7"] and sents == ["Some Title"] and enc.dtype == np.float64
    assert np.array_equal(enc[0].astype(np.float32), v)


def test_sorted_results_ties_lower_index_first():
    s = np.array([[0.5, 0.9, 0.9, 0.1]])
    sc, idx = O.sorted_results(s)
    assert idx.tolist() == [[1, 2, 0, 3]] and sc.tolist() == [[0.9, 0.9, 0.5, 0.1]]


def test_source_encoder_only_gradients_finite_difference():
    """Builder-defined source-encoder-only pair loss (oracle._source_only_gradients) vs central differences."""
    cfg = _cfg(mode="source-encoder-only", V=21, E=4, H=5, S=4)
    cfg["targetSpaceSize"] = 6
    p = {k: v.astype(np.float64) for k, v in O.init_params(cfg, seed=3).items()}
    rng = np.random.RandomState(1)
    src = rng.randint(0, 21, size=(6, 5)).astype(np.int32)
    rows = np.array([0, 2, 2, 5, 1, 3], np.int32)
    z = np.array([1, 0] * 3, np.float64)
    old = O.F32
    try:
        O.F32 = np.float64
        O.FORGET_BIAS, O.L2_EPS, O.LOGIT_SCALE = np.float64(1.0), np.float64(1e-12), np.float64(64.0)

        def f(pp):
            ns = O.encode(pp, cfg, "src", src)
            nt = O.l2_normalize(pp["target_embedding/tgt_seq_embedding"][rows])
            return float(O.loss_and_acc(ns, nt, z)[0])

        loss, _, grads = O.gradients(p, cfg, src, rows, z)
        assert abs(loss - f(p)) < 1e-9 and set(grads) == set(p)
        for name, g in grads.items():
            if isinstance(g, tuple):
                g = O.dense_embedding_grad(g, p[name].shape[0])
            for _ in range(6):
                idx = tuple(rng.randint(0, s) for s in p[name].shape)
                q = {k: v.copy() for k, v in p.items()}
                q[name][idx] += 1e-6
                up = f(q)
                q[name][idx] -= 2e-6
                num = (up - f(q)) / 2e-6
                assert abs(num - g[idx]) < 1e-5 * max(1.0, abs(num)), (name, idx, num, g[idx])
    finally:
        O.F32 = old
        O.FORGET_BIAS, O.L2_EPS, O.LOGIT_SCALE = old(1.0), old(1e-12), old(64.0)


def test_bf16_round_is_round_to_nearest_even():
    x = np.array([1.0, 1.00390625, 1.001953125, 1.005859375, -2.5, 3.0e38, 1e-30, 0.0], np.float32)
    r = O.bf16_round(x)
    assert r[0] == 1.0 and r[1] == np.float32(1.0) and r[2] == np.float32(1.0)      # 1 + 2^-8 ties to even (1.0); 1 + 2^-9 rounds down
    assert r[3] == np.float32(1.0078125)                                             # 1 + 1.5 * 2^-8 rounds up to 1 + 2^-7
    assert np.all((r.view(np.uint32) & 0xFFFF) == 0) and np.all(np.abs(r - x) <= np.abs(x) * 2.0 ** -8)
    torch_r = torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()
    assert np.array_equal(r, torch_r)


def test_topk_fast_equals_topk_including_ties():
    rng = np.random.RandomState(3)
    s = rng.standard_normal((40, 500))
    s[:, 100:130] = s[:, 7:8]                                   # 31 exact ties with column 7 in every row
    s[3, :] = 1.0                                               # a row of nothing but ties: falls back to topk()
    for k in (1, 10, 16):
        a, b = O.topk(s, k), O.topk_fast(s, k)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])
    small = rng.standard_normal((5, 20))
    assert np.array_equal(O.topk(small, 10)[1], O.topk_fast(small, 10)[1])

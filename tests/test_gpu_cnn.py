"""Parity of the HIP text-CNN encoder ('source_only_cnn', sse_model.py:179-211)
and of the free target matrix modes against the CPU oracle (the reference
itself cannot run this mode: tf.concat(3, ...) at :206; the oracle follows the
evident intent, concat on axis 3)."""
import numpy as np
import pytest

from oracle import sse_oracle as O
from tests.util import make_pair, model_params, random_ids

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("V,E,S,T,B", [(300, 50, 512, 64, 37), (100, 30, 64, 80, 9), (60, 8, 16, 5, 3), (200, 50, 64, 33, 70),
                                       (150, 64, 64, 80, 21), (120, 50, 32, 150, 6)])   # 4 sequences per workgroup
def test_cnn_encode_matches_oracle(V, E, S, T, B):
    params = model_params("source_only_cnn", V, E, 96, 96, S, T, N=11)
    m, p = make_pair(params, seed=2)
    ids = random_ids(np.random.RandomState(4), B, T, V, pad_frac=0.5)
    for normalize in (True, False):
        want = O.encode(p, params, "src", ids, normalize=normalize)
        got = m.encode_source(ids, normalize=normalize)
        scale = 1.0 if normalize else max(1.0, float(np.abs(want).max()))
        assert np.abs(got - want).max() <= 1e-4 * scale


@pytest.mark.parametrize("mode", ["source_only_cnn", "source-encoder-only"])
def test_free_target_matrix_is_the_target_encoding(mode):
    """norm_tgt_seq_embedding in these modes = l2_normalize(tgt_seq_embedding) [N,S], whatever is fed
    (sse_model.py:214,233,283)."""
    import sse_amd
    params = model_params(mode, 80, 16, 32, 32, 24, 8, N=13)
    m, p = make_pair(params, seed=1)
    dummy = np.zeros((13, 8), np.int32)
    got = m.encode_target(dummy)
    assert np.abs(got - O.l2_normalize(p["target_embedding/tgt_seq_embedding"])).max() < 1e-6
    assert np.array_equal(m.encode_target(dummy, normalize=False), p["target_embedding/tgt_seq_embedding"])
    with pytest.raises(sse_amd.SSEError):
        m.encode_target(np.zeros((5, 8), np.int32))


def _cnn_batch(rng, B, T, V, N, pad_frac=0.5):
    src = np.repeat(random_ids(rng, B // 2, T, V, pad_frac), 2, axis=0)       # pos,neg rows share the source
    rows = rng.randint(0, N, size=B).astype(np.int32)
    return src, rows, np.tile(np.array([1.0, 0.0], np.float32), B // 2)


@pytest.mark.parametrize("V,E,S,T,B,N", [(300, 50, 512, 64, 48, 571), (90, 24, 64, 20, 10, 17), (60, 8, 16, 5, 4, 5),
                                         (200, 50, 64, 80, 26, 33)])     # reference default T=80: 4-sequence tiles
def test_cnn_train_step_matches_oracle(V, E, S, T, B, N):
    """BUILDER-DEFINED CNN pair loss (BASELINE configs[4]; oracle._cnn_gradients): one step of forward with arg-max
    tape, gather/scatter backward, clip, Adagrad vs the oracle.  Tolerances as for the LSTM step; a near-tie of two
    pooled positions (|dv| ~ 1e-7) may legitimately route one filter's gradient elsewhere, hence 5e-4."""
    params = model_params("source_only_cnn", V, E, 96, 96, S, T, N=N, lr=0.9)
    m, p = make_pair(params, seed=6)
    st = O.new_optimizer_state(p)
    src, rows, z = _cnn_batch(np.random.RandomState(3), B, T, V, N)
    want_loss, want_acc = O.train_step(p, st, params, src, rows, z, 0.9)
    loss, acc = m.train_step(src, rows, z)
    assert loss == pytest.approx(float(want_loss), rel=1e-5, abs=1e-6)
    assert acc == pytest.approx(float(want_acc), abs=1e-6)
    got = m.get_variables(with_slots=True)
    for name, w in p.items():
        assert np.abs(got[name].reshape(w.shape) - w).max() < 5e-4, name
        assert np.abs(got[name + "/Adagrad"].reshape(w.shape) - st[name]).max() < 5e-4, name + "/Adagrad"
    assert m.handle.global_step == 1
    # the encoder sees the updated weights (layouts refreshed)
    ids = random_ids(np.random.RandomState(1), 5, T, V, 0.3)
    assert np.abs(m.encode_source(ids) - O.encode(p, params, "src", ids)).max() < 2e-3


def test_cnn_training_learns_and_rejects_bad_rows():
    import sse_amd
    V, E, S, T, N = 120, 16, 32, 12, 9
    params = model_params("source_only_cnn", V, E, 96, 96, S, T, N=N, lr=0.1)
    m, p = make_pair(params, seed=1)
    m.handle.learning_rate = 0.1
    st = O.new_optimizer_state(p)
    src, rows, z = _cnn_batch(np.random.RandomState(5), 24, T, V, N)
    got, want = [], []
    for _ in range(12):
        want.append(float(O.train_step(p, st, params, src, rows, z, 0.1)[0]))
        got.append(m.train_step(src, rows, z)[0])
    assert min(got) < got[0]
    assert np.allclose(got, want, rtol=5e-3, atol=1e-4)
    bad = rows.copy()
    bad[3] = N
    with pytest.raises(sse_amd.SSEError):
        m.train_step(src, bad, z)
    with pytest.raises(ValueError):
        m.train_step(src, np.zeros_like(src), z)                   # token ids where target rows are expected


def test_cnn_data_parallel_two_logical_ranks():
    import torch
    V, E, S, T, N, B = 150, 20, 48, 16, 31, 40
    params = model_params("source_only_cnn", V, E, 96, 96, S, T, N=N, lr=0.9)
    (m0, p), (m1, _) = make_pair(params, seed=3), make_pair(params, seed=3)
    st = O.new_optimizer_state(p)
    src, rows, z = _cnn_batch(np.random.RandomState(9), B, T, V, N)
    want = O.train_step(p, st, params, src, rows, z, 0.9)
    arenas = []
    for m, sl in ((m0, slice(0, 26)), (m1, slice(26, B))):       # uneven split
        a = torch.zeros(m.handle.train_grad_count(), dtype=torch.float32, device="cuda:0")
        m.handle.train_bind_arena(a)
        m.handle.train_grads(src[sl], rows[sl], z[sl], rows_global=B)
        arenas.append(a)
    torch.cuda.synchronize()
    total = arenas[0] + arenas[1]
    for a in arenas:
        a.copy_(total)
    torch.cuda.synchronize()
    res = [m.handle.train_apply() for m in (m0, m1)]
    assert res[0] == res[1]
    assert res[0][0] == pytest.approx(float(want[0]), rel=1e-5)
    g0, g1 = m0.get_variables(), m1.get_variables()
    for name, w in p.items():
        assert np.array_equal(g0[name], g1[name]), name
        assert np.abs(g0[name].reshape(w.shape) - w).max() < 5e-4, name


@pytest.mark.parametrize("V,E,S,T,B", [(300, 50, 512, 64, 37), (100, 30, 64, 80, 9), (60, 8, 16, 5, 3), (150, 64, 64, 80, 21)])
def test_cnn_bf16_storage_variant_matches_bf16_oracle(V, E, S, T, B):
    """Option cnn_bf16 (BASELINE configs[4] names bf16; builder-defined precision): embeddings and filters rounded
    to bf16, fp32 accumulation -> equals the oracle with the same rounding to fp32 accumulation error, and stays
    within bf16 storage error of the fp32 encoding."""
    params = model_params("source_only_cnn", V, E, 96, 96, S, T, N=11)
    m, p = make_pair(params, seed=2)
    ids = random_ids(np.random.RandomState(4), B, T, V, pad_frac=0.5)
    exact = m.encode_source(ids)
    m.handle.set_option("cnn_bf16", 1)
    for normalize in (True, False):
        want = O.encode(p, params, "src", ids, normalize=normalize, cnn_bf16=True)
        got = m.encode_source(ids, normalize=normalize)
        scale = 1.0 if normalize else max(1.0, float(np.abs(want).max()))
        assert np.abs(got - want).max() <= 1e-4 * scale
    got = m.encode_source(ids)
    assert 0 < np.abs(got - exact).max() < 3e-2                      # genuinely a different precision, and a close one
    cos = np.sum(got * exact, axis=1)
    assert cos.min() > 0.9995
    m.handle.set_option("cnn_bf16", 0)
    assert np.array_equal(m.encode_source(ids), exact)               # back to the exact fp32 path


@pytest.mark.parametrize("V,E,S,T,B,N", [(300, 50, 512, 64, 48, 571), (90, 24, 64, 20, 10, 17), (200, 50, 64, 80, 26, 33)])
def test_cnn_bf16_train_step_matches_bf16_oracle(V, E, S, T, B, N):
    """BASELINE configs[4] ("bf16 ... train"): with option cnn_bf16 the training forward runs the convolution on the
    bf16 matrix pipe over bf16-rounded embeddings / filters (fp32 masters and accumulation), the backward routes
    through the rounded operands (oracle._cnn_gradients(bf16=True)); three steps, so the refreshed bf16 copies of
    the UPDATED masters are what steps 2 and 3 read."""
    params = model_params("source_only_cnn", V, E, 96, 96, S, T, N=N, lr=0.9)
    cfg16 = dict(params, cnn_bf16=True)
    m, p = make_pair(params, seed=6)
    m.handle.set_option("cnn_bf16", 1)
    st = O.new_optimizer_state(p)
    src, rows, z = _cnn_batch(np.random.RandomState(3), B, T, V, N)
    p32 = {k: v.copy() for k, v in p.items()}
    loss32 = float(O.train_step(p32, O.new_optimizer_state(p32), params, src, rows, z, 0.9)[0])
    for step in range(3):
        want_loss, want_acc = O.train_step(p, st, cfg16, src, rows, z, 0.9)
        loss, acc = m.train_step(src, rows, z)
        assert loss == pytest.approx(float(want_loss), rel=2e-5, abs=2e-6), step
        assert acc == pytest.approx(float(want_acc), abs=1e-6), step
        if step == 0:
            assert loss != pytest.approx(loss32, rel=1e-7)          # genuinely the bf16 function
    got = m.get_variables(with_slots=True)
    for name, w in p.items():
        assert np.abs(got[name].reshape(w.shape) - w).max() < 1e-3, name
        assert np.abs(got[name + "/Adagrad"].reshape(w.shape) - st[name]).max() < 1e-3, name + "/Adagrad"
    # masters stay float32: the updated filters are not bf16-representable
    w = got["source_only_cnn/conv-maxpool-3/W"]
    assert not np.array_equal(O.bf16_round(w), w)
    ids = random_ids(np.random.RandomState(1), 5, T, V, 0.3)
    assert np.abs(m.encode_source(ids) - O.encode(p, params, "src", ids, cnn_bf16=True)).max() < 2e-3


def test_cnn_bf16_option_is_rejected_outside_cnn_mode():
    import sse_amd
    m, _ = make_pair(model_params("dual-encoder", 50, 8, 16, 16, 8, 5), seed=0)
    with pytest.raises(sse_amd.SSEError):
        m.handle.set_option("cnn_bf16", 1)

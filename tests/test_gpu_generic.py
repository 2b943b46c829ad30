"""The any-shape LSTM path (csrc/lstm_generic.hip) against the CPU oracle: shapes the fused kernels are not laid out for
(cell size > 512 for inference, > 256 for training; embedding_size > 64; encoding_size > 512) used to be rejected -- the
reference builds its graph for any --src_cell_size / --embedding_size / --encoding_size (sse_train.py:60-74,
sse_model.py:113-126,236-275).  Same arithmetic, same tolerances as the fused path's tests; on shapes both paths accept
(option train_generic) the two are also compared with each other."""
import numpy as np
import pytest

from oracle import sse_oracle as O
from tests.util import LOSS_REL_EXACT, make_pair, model_params, random_ids

pytestmark = pytest.mark.gpu


def _batch(rng, B, T, V, pad_frac=0.6):
    src = np.repeat(random_ids(rng, B // 2, T, V, pad_frac), 2, axis=0)
    tgt = random_ids(rng, B, T, V, pad_frac)
    return src, tgt, np.tile(np.array([1.0, 0.0], np.float32), B // 2)


def _check_step(m, p, params, st, src, tgt, z, lr=0.9, steps=2):
    for _ in range(steps):
        want = O.train_step(p, st, params, src, tgt, z, lr)
        got = m.train_step(src, tgt, z)
        assert got[0] == pytest.approx(float(want[0]), rel=LOSS_REL_EXACT, abs=1e-6)
        assert got[1] == pytest.approx(float(want[1]), abs=1e-6)
    v = m.get_variables(with_slots=True)
    for name, w in p.items():
        assert np.abs(v[name].reshape(w.shape) - w).max() < 2e-4, name
        assert np.abs(v[name + "/Adagrad"].reshape(w.shape) - st[name]).max() < 2e-4, name + "/Adagrad"


@pytest.mark.parametrize("mode,V,E,Hs,Ht,S,T,B", [
    ("dual-encoder", 300, 50, 96, 64, 64, 12, 32),       # shapes the fused kernels take too: forced onto the generic path
    ("shared-encoder", 200, 40, 96, 96, 50, 9, 20),
    ("dual-encoder", 150, 7, 33, 129, 19, 5, 6),          # odd sizes: cell sizes not multiples of 8, S not a multiple of 8
])
def test_generic_train_step_matches_oracle_and_fused_path(mode, V, E, Hs, Ht, S, T, B):
    params = model_params(mode, V, E, Hs, Ht, S, T, lr=0.9)
    m, p = make_pair(params, seed=3)
    fused, _ = make_pair(params, seed=3)
    m.handle.set_option("train_generic", 1)
    st = O.new_optimizer_state(p)
    src, tgt, z = _batch(np.random.RandomState(11), B, T, V)
    _check_step(m, p, params, st, src, tgt, z)
    for _ in range(2):
        fused.train_step(src, tgt, z)
    a, b = m.get_variables(), fused.get_variables()
    for k in a:
        assert np.abs(a[k] - b[k]).max() < 2e-4, k


@pytest.mark.parametrize("mode,V,E,Hs,Ht,S,T,B", [
    ("dual-encoder", 200, 50, 300, 300, 64, 10, 32),      # --src_cell_size=300: training above the fused kernels' 256
    ("shared-encoder", 200, 50, 512, 512, 128, 6, 16),    # cell size 512 (fused inference, generic training)
    ("dual-encoder", 120, 100, 128, 96, 64, 8, 24),       # --embedding_size=100 > 64
    ("dual-encoder", 100, 128, 600, 96, 72, 5, 8),        # source cell size 600 > 512: generic inference AND training, mixed with a fused-size target
    ("source-encoder-only", 90, 70, 64, 64, 40, 6, 10),   # builder-defined mode (free target matrix), E > 64
    ("dual-encoder", 90, 70, 520, 7, 16, 5, 6),           # cell sizes whose padded width is not a multiple of 32 (520; 7 -> 8): found by
    ("shared-encoder", 300, 100, 5, 5, 256, 6, 128),      # tools/fuzz_parity.py -- the dh GEMM's grid rounded the tile count DOWN
])
def test_shapes_outside_the_fused_kernels_train_and_encode(mode, V, E, Hs, Ht, S, T, B):
    params = model_params(mode, V, E, Hs, Ht, S, T, N=13, lr=0.9)
    m, p = make_pair(params, seed=5)
    st = O.new_optimizer_state(p)
    rng = np.random.RandomState(2)
    src, tgt, z = _batch(rng, B, T, V)
    if mode == "source-encoder-only":
        tgt = rng.randint(0, 13, size=B).astype(np.int32)
    _check_step(m, p, params, st, src, tgt, z)
    ids = random_ids(rng, 70, T, V, 0.4)
    for side, name in ((0, "src"), (1, "tgt")):
        if mode == "source-encoder-only" and side == 1:
            continue
        for normalize in (True, False):
            want = O.encode(p, params, name, ids, normalize=normalize)
            got = (m.encode_source if side == 0 else m.encode_target)(ids, normalize=normalize)
            scale = 1.0 if normalize else max(1.0, float(np.abs(want).max()))
            assert np.abs(got - want).max() <= 1e-4 * scale, (name, normalize)


def test_wide_encoding_and_big_cell_inference_rows_in_chunks():
    """encoding_size 600 > 512 and cell size 700: inference only through the generic path, more rows than one chunk's 32-row
    padding, ids out of range still raise."""
    import sse_amd
    params = model_params("dual-encoder", 80, 20, 700, 40, 600, 7)
    m, p = make_pair(params, seed=9)
    ids = random_ids(np.random.RandomState(4), 133, 7, 80, 0.5)
    want = O.encode(p, params, "src", ids)
    got = m.encode_source(ids)
    assert got.shape == (133, 600) and np.abs(got - want).max() < 1e-4
    got_t = m.encode_target(ids[:5])                                      # the small target cell: generic too (S > 512)
    assert np.abs(got_t - O.encode(p, params, "tgt", ids[:5])).max() < 1e-4
    bad = ids.copy()
    bad[7, 3] = 80
    with pytest.raises(sse_amd.SSEError):
        m.encode_source(bad)
    assert np.abs(m.encode_source(ids) - want).max() < 1e-4             # the handle stays usable


def test_generic_path_caches_its_packs_and_cancels_a_step_with_a_bad_id():
    """the any-shape path rebuilt its packed weights with every encode and forced a host round trip in every train
    step.  Now the packs are cached against the handle's weight version (an update or sse_set_variable invalidates them) and
    gen_dx_scatter_kernel validates the ids itself: a token id out of range raises through the deferred check, the update is
    cancelled on the device (weights AND Adagrad slots unchanged), the handle stays usable."""
    import sse_amd
    params = model_params("dual-encoder", 90, 70, 300, 300, 64, 6, lr=0.9)   # E = 70 > 64, H = 300: generic for training and inference
    m, p = make_pair(params, seed=3)
    st = O.new_optimizer_state(p)
    rng = np.random.RandomState(8)
    src, tgt, z = _batch(rng, 16, 6, 90)
    ids = random_ids(rng, 40, 6, 90, 0.3)
    first = m.encode_source(ids)
    assert np.array_equal(m.encode_source(ids), first)                      # second call: cached packs, same bits
    assert np.abs(first - O.encode(p, params, "src", ids)).max() < 1e-4
    before = m.get_variables(with_slots=True)
    bad = src.copy()
    bad[3, 2] = 90
    with pytest.raises(sse_amd.SSEError, match="out of range"):
        m.train_step(bad, tgt, z)
    after = m.get_variables(with_slots=True)
    for k in before:
        assert np.array_equal(before[k], after[k]), k
    _check_step(m, p, params, st, src, tgt, z)                               # a good step afterwards: matches the oracle ...
    got = m.encode_source(ids)                                               # ... and the encode sees the UPDATED weights
    assert np.abs(got - O.encode(p, params, "src", ids)).max() < 1e-4
    assert not np.array_equal(got, first)
    p2 = {k: (v * 0.5).astype(np.float32) for k, v in p.items()}
    m.set_variables(p2)                                                      # sse_set_variable invalidates the packs too
    assert np.abs(m.encode_source(ids) - O.encode(p2, params, "src", ids)).max() < 1e-4

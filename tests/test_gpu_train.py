"""Parity of the HIP training step (loss, BPTT, clip_by_global_norm, Adagrad;
through the C ABI) against the CPU oracle's restatement of
sse_model.py:279-302,355-364.  Tolerances: loss/acc 1e-5 relative; updated
variables and Adagrad slots 2e-4 absolute after one step (fp32, different
summation orders), drift stays below 2e-3 after several steps."""
import numpy as np
import pytest

from oracle import sse_oracle as O
from tests.util import (LOSS_REL, LOSS_REL_EXACT, LOSS_REL_SPLIT, exact_fp32_training, make_pair, model_params, random_ids,
                        split_bf16_training)

pytestmark = pytest.mark.gpu


def _batch(rng, B, T, V, pad_frac=0.6):
    src = random_ids(rng, B // 2, T, V, pad_frac)
    src = np.repeat(src, 2, axis=0)                       # data.py:95-115: each source appears twice (pos, neg)
    tgt = random_ids(rng, B, T, V, pad_frac)
    labels = np.tile(np.array([1.0, 0.0], np.float32), B // 2)
    return src, tgt, labels


CASES = [
    ("dual-encoder", 400, 50, 256, 256, 256, 32, 128),
    ("shared-encoder", 300, 40, 96, 96, 50, 50, 64),
    ("dual-encoder", 90, 30, 64, 128, 64, 6, 10),
    ("shared-encoder", 60, 8, 32, 32, 16, 3, 2),
]


@pytest.mark.parametrize("split", [False, True])       # default: fp32 MFMA throughout (the reference's arithmetic) / opt-in split-bf16 GEMMs
@pytest.mark.parametrize("mode,V,E,Hs,Ht,S,T,B", CASES)
def test_one_train_step_matches_oracle(mode, V, E, Hs, Ht, S, T, B, split):
    params = model_params(mode, V, E, Hs, Ht, S, T, lr=0.9)
    m, p = make_pair(params, seed=3)
    if split:
        split_bf16_training(m)
    st = O.new_optimizer_state(p)
    rng = np.random.RandomState(11)
    src, tgt, z = _batch(rng, B, T, V)
    want_loss, want_acc = O.train_step(p, st, params, src, tgt, z, 0.9)
    loss, acc = m.train_step(src, tgt, z)
    assert loss == pytest.approx(float(want_loss), rel=LOSS_REL_SPLIT if split else LOSS_REL_EXACT, abs=1e-6)
    assert acc == pytest.approx(float(want_acc), abs=1e-6)
    got = m.get_variables(with_slots=True)
    for name, w in p.items():
        assert np.abs(got[name].reshape(w.shape) - w).max() < 2e-4, name
        assert np.abs(got[name + "/Adagrad"].reshape(w.shape) - st[name]).max() < 2e-4, name + "/Adagrad"
    assert m.handle.global_step == 1


@pytest.mark.parametrize("mode,V,E,Hs,Ht,S,T,B,gen1", [
    ("dual-encoder", 400, 50, 256, 256, 256, 32, 128, 1),       # option train_gen1: lstm_bwd_kernel + dx_kernel + bias partials
    ("shared-encoder", 300, 40, 96, 96, 50, 50, 64, 1),
    ("dual-encoder", 300, 64, 128, 256, 64, 9, 128, 0),          # E = 64: no room for the constant-1 column -> first generation by itself
    ("dual-encoder", 300, 63, 128, 256, 64, 9, 128, 0),          # E = 63: widest embedding of the second generation (x tile single-buffered)
    ("dual-encoder", 300, 32, 96, 64, 40, 7, 66, 0),             # E = 32: the constant-1 column opens the second x tile
    ("shared-encoder", 200, 7, 40, 40, 24, 5, 130, 0),           # one live e-tile, cells far below the padded size
])
def test_fp32_train_step_kernel_generations(mode, V, E, Hs, Ht, S, T, B, gen1):
    """The fp32 train step exists twice: lstm_fwd_kernel<TSW> + lstm_bwd2_kernel + d(bias) from the weight-gradient GEMM
    (default), and the first-generation kernels (option train_gen1; taken by themselves for E = 64).  Both against the
    oracle at the exact-path tolerance, two steps."""
    params = model_params(mode, V, E, Hs, Ht, S, T, lr=0.9)
    m, p = make_pair(params, seed=13)
    m.handle.set_option("train_gen1", gen1)
    st = O.new_optimizer_state(p)
    rng = np.random.RandomState(5)
    for step in range(2):
        src, tgt, z = _batch(rng, B, T, V)
        want = O.train_step(p, st, params, src, tgt, z, 0.9)
        got = m.train_step(src, tgt, z)
        assert got[0] == pytest.approx(float(want[0]), rel=LOSS_REL_EXACT, abs=1e-6)
        assert got[1] == pytest.approx(float(want[1]), abs=1e-6)
    v = m.get_variables(with_slots=True)
    for name, w in p.items():
        assert np.abs(v[name].reshape(w.shape) - w).max() < 2e-4, name
        assert np.abs(v[name + "/Adagrad"].reshape(w.shape) - st[name]).max() < 2e-4, name


def test_several_steps_track_oracle_and_loss_falls():
    params = model_params("dual-encoder", 200, 50, 96, 96, 64, 12, lr=0.05)
    m, p = make_pair(params, seed=5)
    m.handle.learning_rate = 0.05
    st = O.new_optimizer_state(p)
    rng = np.random.RandomState(2)
    src, tgt, z = _batch(rng, 32, 12, 200)
    got_losses, want_losses = [], []
    for _ in range(10):
        want_losses.append(float(O.train_step(p, st, params, src, tgt, z, 0.05)[0]))
        got_losses.append(m.train_step(src, tgt, z)[0])
    assert min(got_losses) < got_losses[0]           # Adagrad at these rates is not monotone; parity is the bar
    assert np.allclose(got_losses, want_losses, rtol=2e-3, atol=2e-4)
    got = m.get_variables()
    for name, w in p.items():
        assert np.abs(got[name].reshape(w.shape) - w).max() < 2e-3, name
    # encoders see the updated weights (re-layout after the step)
    ids = random_ids(rng, 5, 12, 200)
    assert np.abs(m.encode_source(ids) - O.encode(p, params, "src", ids)).max() < 1e-3


def test_clip_engages_and_untouched_embedding_rows_stay():
    params = model_params("dual-encoder", 500, 20, 32, 32, 16, 5, lr=0.9)
    m, p = make_pair(params, seed=7)
    for k in p:                                            # large weights -> large gradients -> clipping active
        if k.endswith("_M"):
            p[k] = (p[k] * 30).astype(np.float32)
    m.set_variables(p)
    st = O.new_optimizer_state(p)
    rng = np.random.RandomState(1)
    src, tgt, z = _batch(rng, 16, 5, 40)                   # only ids < 40 are touched
    _, _, grads = O.gradients(p, params, src, tgt, z)
    assert O.global_norm(grads) > 5.0
    before = p["word_embedding"].copy()
    O.train_step(p, st, params, src, tgt, z, 0.9)
    m.train_step(src, tgt, z)
    got = m.get_variables(with_slots=True)
    assert np.array_equal(got["word_embedding"][40:], before[40:])
    assert np.all(got["word_embedding/Adagrad"][40:] == np.float32(0.1))
    for name, w in p.items():
        assert np.abs(got[name].reshape(w.shape) - w).max() < 5e-4, name


def test_session_run_train_contract_and_lr_decay():
    import sse_amd
    params = model_params("shared-encoder", 80, 16, 32, 32, 24, 6, lr=0.9)
    m, p = make_pair(params)
    sess = sse_amd.Session(m)
    rng = np.random.RandomState(0)
    src, tgt, z = _batch(rng, 8, 6, 80)
    d = m.get_train_feed_dict(src.tolist(), tgt.tolist(), z.tolist())
    _, summary, step_loss, step_acc = sess.run([m.train, m.add_summaries(), m.loss, m.train_acc], feed_dict=d)
    want_loss, want_acc = O.train_step(p, O.new_optimizer_state(p), params, src, tgt, z, 0.9)
    assert step_loss == pytest.approx(float(want_loss), rel=LOSS_REL)
    assert m.global_step.eval() == 1 and m.learning_rate.eval() == pytest.approx(0.9)
    sess.run(m.learning_rate_decay_op)
    assert m.learning_rate.eval() == pytest.approx(float(O.decayed_learning_rate(0.9, 0.99)))


def test_checkpoint_roundtrip(tmp_path):
    import sse_amd
    params = model_params("dual-encoder", 60, 8, 32, 32, 16, 4)
    m, p = make_pair(params)
    rng = np.random.RandomState(0)
    src, tgt, z = _batch(rng, 4, 4, 60)
    m.train_step(src, tgt, z)
    path = m.save(None, str(tmp_path / "SSE-LSTM.ckpt-BestEver"))
    assert sse_amd.get_checkpoint_state(str(tmp_path)) == path
    m2 = sse_amd.SSEModel(params)
    m2.saver.restore(None, path)
    a, b = m.get_variables(with_slots=True), m2.get_variables(with_slots=True)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert m2.handle.global_step == 1
    ids = random_ids(rng, 3, 4, 60)
    assert np.array_equal(m.encode_target(ids), m2.encode_target(ids))


def test_data_parallel_two_logical_ranks():
    """SURVEY 8e "Training": grads on two half batches (two handles = two logical ranks on one GPU), the flat gradient
    arenas summed (what the RCCL all-reduce does), apply on both -> same loss/acc/weights as the single-process step on
    the whole batch (oracle and HIP), and the two ranks end bit-identical."""
    import torch
    import sse_amd
    params = model_params("dual-encoder", 300, 50, 128, 128, 64, 10, lr=0.9)
    rng = np.random.RandomState(4)
    src, tgt, z = _batch(rng, 96, 10, 300)
    (m0, p), (m1, _), (mf, _) = make_pair(params, seed=7), make_pair(params, seed=7), make_pair(params, seed=7)
    st = O.new_optimizer_state(p)
    want = O.train_step(p, st, params, src, tgt, z, 0.9)
    full = mf.train_step(src, tgt, z)
    arenas = []
    for r, m in enumerate((m0, m1)):
        n = m.handle.train_grad_count()
        a = torch.zeros(n, dtype=torch.float32, device="cuda:0")
        m.handle.train_bind_arena(a)
        s, t, l = sse_amd.split_batch(src, tgt, z, r, 2) if r == 0 else (src[48:], tgt[48:], z[48:])
        m.handle.train_grads(s, t, l, rows_global=96)
        arenas.append(a)
    torch.cuda.synchronize()
    total = arenas[0] + arenas[1]
    assert float(total[-1]) == 96.0                                # rows
    for a in arenas:
        a.copy_(total)
    torch.cuda.synchronize()
    res = [m.handle.train_apply() for m in (m0, m1)]
    assert res[0] == res[1]
    assert res[0][0] == pytest.approx(float(want[0]), rel=LOSS_REL) and res[0][1] == pytest.approx(float(want[1]), abs=1e-6)
    assert res[0][0] == pytest.approx(full[0], rel=1e-5)
    g0, g1, gf = (m.get_variables(with_slots=True) for m in (m0, m1, mf))
    for name, w in p.items():
        assert np.array_equal(g0[name], g1[name]), name
        assert np.abs(g0[name].reshape(w.shape) - w).max() < 2e-4, name
        assert np.abs(g0[name] - gf[name]).max() < 2e-5, name
        assert np.abs(g0[name + "/Adagrad"].reshape(w.shape) - st[name]).max() < 2e-4, name
    assert m0.handle.global_step == 1


def test_packed_embedding_gradient_exchange_three_logical_ranks():
    """sse_train_pack_embedding_grad / sse_train_unpack_embedding_grad (the embedding gradient as (row id,
    gradient row) pairs): three handles = three logical ranks on one GPU.  The packed buffers laid back to back (what the
    all-gather delivers) and unpacked give, on every rank, exactly the sum of the three dense blocks in rank order; an
    undersized buffer raises error bit 8 and cancels the update."""
    import torch
    import sse_amd
    V, E, T = 3000, 50, 10
    params = model_params("dual-encoder", V, E, 128, 128, 64, T, lr=0.9)
    rng = np.random.RandomState(14)
    src, tgt, z = _batch(rng, 90, T, V)
    ms = [make_pair(params, seed=7)[0] for _ in range(3)]
    arenas, dense = [], []
    for r, m in enumerate(ms):
        a = torch.zeros(m.handle.train_grad_count(), dtype=torch.float32, device="cuda:0")
        m.handle.train_bind_arena(a)
        m.handle.train_grads(*sse_amd.split_batch(src, tgt, z, r, 3), rows_global=90)
        arenas.append(a)
    torch.cuda.synchronize()
    dense = [a[:V * E].clone() for a in arenas]
    cap = min(V, 2 * 30 * T)
    n = ms[0].handle.dp_packed_floats(cap)
    assert n == 4 + ((cap + 3) & ~3) + cap * E
    gathered = torch.zeros(3 * n, dtype=torch.float32, device="cuda:0")
    for r, m in enumerate(ms):
        m.handle.dp_pack_embedding(cap, gathered[r * n:(r + 1) * n])
    torch.cuda.synchronize()
    counts = [int(gathered[r * n:r * n + 1].view(torch.int32)[0]) for r in range(3)]
    for r in range(3):
        touched = int((dense[r].view(V, E) != 0).any(dim=1).sum())
        assert counts[r] == touched and 0 < touched <= cap
        ids = gathered[r * n + 4:r * n + 4 + counts[r]].view(torch.int32).cpu().numpy()
        assert len(set(ids.tolist())) == counts[r]                 # every touched row once
    want = (dense[0] + dense[1]) + dense[2]                        # rank order
    for m, a in zip(ms, arenas):
        m.handle.dp_unpack_embedding(gathered, 3, cap)
    torch.cuda.synchronize()
    for a in arenas:
        assert torch.equal(a[:V * E], want)
    # the rest of the arena is summed as the dense exchange does it; the three ranks then apply the same update
    tail = sum(a[V * E:] for a in arenas)
    for a in arenas:
        a[V * E:] = tail
    res = [m.handle.train_apply() for m in ms]
    assert res[0] == res[1] == res[2]
    g = [m.get_variables(with_slots=True) for m in ms]
    assert all(np.array_equal(g[0][k], g[1][k]) and np.array_equal(g[0][k], g[2][k]) for k in g[0])
    # undersized buffer: flagged on the device, the update is cancelled and the error names the exchange
    m = ms[0]
    before = m.get_variables()
    m.handle.train_grads(*sse_amd.split_batch(src, tgt, z, 0, 3), rows_global=90)
    small = torch.zeros(m.handle.dp_packed_floats(8), dtype=torch.float32, device="cuda:0")
    m.handle.dp_pack_embedding(8, small)
    with pytest.raises(sse_amd.SSEError, match="embedding-gradient exchange"):
        m.handle.train_apply()
    after = m.get_variables()
    assert all(np.array_equal(before[k], after[k]) for k in before)
    assert int(small[:1].view(torch.int32)[0]) > 8                 # the counter kept counting; nothing past slot 8 was written


def test_data_parallel_trainer_world1_and_arena_ownership():
    import sse_amd
    params = model_params("shared-encoder", 80, 16, 32, 32, 16, 5, lr=0.5)
    rng = np.random.RandomState(8)
    src, tgt, z = _batch(rng, 12, 5, 80)
    (ma, _), (mb, _) = make_pair(params, seed=2), make_pair(params, seed=2)
    tr = sse_amd.DataParallelTrainer(ma.handle, device="cuda:0")
    got = [tr.train_step(src, tgt, z) for _ in range(2)]
    want = [mb.train_step(src, tgt, z) for _ in range(2)]
    # same kernels either way; the embedding gradient is a float atomicAdd scatter, so runs agree to rounding only
    assert np.allclose(got, want, rtol=1e-6, atol=1e-7)
    va, vb = ma.get_variables(with_slots=True), mb.get_variables(with_slots=True)
    assert all(np.abs(va[k] - vb[k]).max() < 1e-6 for k in va)
    ma.handle.train_set_grad_arena(None, 0)                        # back to a library-owned arena
    assert np.allclose(ma.train_step(src, tgt, z), mb.train_step(src, tgt, z), rtol=1e-6, atol=1e-7)
    with pytest.raises(sse_amd.SSEError):
        ma.handle.train_apply()                                    # nothing pending
    with pytest.raises(sse_amd.SSEError):
        ma.handle.train_set_grad_arena(tr.arena.data_ptr(), 5)     # wrong size


@pytest.mark.parametrize("mode,H", [("dual-encoder", 256), ("shared-encoder", 128), ("source-encoder-only", 96)])
def test_paired_batch_runs_the_source_encoder_once_per_pair(mode, H):
    """data.py:95-115 batches: rows 2i and 2i+1 share their source sequence.  With B % 128 == 0 the source forward and
    the source side of the dK GEMM run on B/2 rows (option train_pair_dedup, default on); same step as the per-row
    path (option off) up to fp32 summation order, and as the oracle."""
    V, E, S, T, B, N = 300, 50, 64, 12, 256, 23
    params = model_params(mode, V, E, H, H, S, T, N=N, lr=0.9)
    (ma, p), (mb, _) = make_pair(params, seed=11), make_pair(params, seed=11)
    mb.handle.set_option("train_pair_dedup", 0)
    st = O.new_optimizer_state(p)
    rng = np.random.RandomState(5)
    table = mode == "source-encoder-only"
    for step in range(2):
        src = np.repeat(random_ids(rng, B // 2, T, V, 0.4), 2, axis=0)
        tgt = rng.randint(0, N, size=B).astype(np.int32) if table else random_ids(rng, B, T, V, 0.4)
        z = np.tile(np.array([1.0, 0.0], np.float32), B // 2)
        want = O.train_step(p, st, params, src, tgt, z, 0.9)
        la, lb = ma.train_step(src, tgt, z), mb.train_step(src, tgt, z)
        assert la[0] == pytest.approx(lb[0], rel=2e-6) and la[1] == pytest.approx(lb[1], abs=1e-6)
        assert la[0] == pytest.approx(float(want[0]), rel=LOSS_REL, abs=1e-6)
    va, vb = ma.get_variables(with_slots=True), mb.get_variables(with_slots=True)
    for name, w in p.items():
        assert np.abs(va[name] - vb[name]).max() < 2e-5, name
        assert np.abs(va[name].reshape(w.shape) - w).max() < 2e-4, name
        assert np.abs(va[name + "/Adagrad"].reshape(w.shape) - st[name]).max() < 2e-4, name
    # an unpaired batch of the same size takes the per-row path on both handles: identical results
    src = random_ids(rng, B, T, V, 0.4)
    tgt = rng.randint(0, N, size=B).astype(np.int32) if table else random_ids(rng, B, T, V, 0.4)
    z = np.tile(np.array([1.0, 0.0], np.float32), B // 2)
    la, lb = ma.train_step(src, tgt, z), mb.train_step(src, tgt, z)
    assert la[0] == pytest.approx(lb[0], rel=1e-4)


@pytest.mark.parametrize("fwd,bwd,dk", [(0, 0, 0), (1, 0, 1), (0, 1, 1), (0, 0, 1), (1, 1, 1)])
def test_split_operand_options_are_independent(fwd, bwd, dk):
    """train_fwd_x3 / train_bwd_x3 / train_dk_x3 select the split-bf16 GEMMs kernel by kernel (the forward and BPTT
    variants need the split tape / dG formats of train_dk_x3); every combination is the same step to the loss tolerance
    of its forward and 2e-4 on the updated weights."""
    params = model_params("dual-encoder", 300, 50, 256, 96, 64, 9, lr=0.9)
    m, p = make_pair(params, seed=17)
    for name, v in (("train_fwd_x3", fwd), ("train_bwd_x3", bwd), ("train_dk_x3", dk)):
        m.handle.set_option(name, v)
    st = O.new_optimizer_state(p)
    rng = np.random.RandomState(3)
    for step in range(2):
        src, tgt, z = _batch(rng, 128, 9, 300)
        want = O.train_step(p, st, params, src, tgt, z, 0.9)
        got = m.train_step(src, tgt, z)
        assert got[0] == pytest.approx(float(want[0]), rel=LOSS_REL_SPLIT if fwd else LOSS_REL_EXACT, abs=1e-6)
    v = m.get_variables(with_slots=True)
    for name, w in p.items():
        assert np.abs(v[name].reshape(w.shape) - w).max() < 2e-4, name
        assert np.abs(v[name + "/Adagrad"].reshape(w.shape) - st[name]).max() < 2e-4, name


def test_handles_release_their_device_memory():
    """Create / train / encode / destroy repeatedly: free device memory does not drift (scratch, tapes, arena)."""
    import gc
    import torch
    params = model_params("dual-encoder", 500, 50, 128, 128, 64, 16, lr=0.5)
    cparams = model_params("source_only_cnn", 500, 50, 96, 96, 64, 16, N=9, lr=0.5)
    rng = np.random.RandomState(0)
    src, tgt, z = _batch(rng, 256, 16, 500)
    rows = rng.randint(0, 9, size=256).astype(np.int32)

    def cycle():
        m, _ = make_pair(params, seed=1)
        m.train_step(src, tgt, z)
        m.encode_source(src)
        m.handle.close()
        c, _ = make_pair(cparams, seed=1)
        c.train_step(src, rows, z)
        c.handle.close()
        del m, c
        gc.collect()

    cycle()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(8):
        cycle()
    torch.cuda.synchronize()
    assert free0 - torch.cuda.mem_get_info()[0] < 8 << 20


@pytest.mark.parametrize("V,E,H,S,T,B,N", [(200, 50, 128, 64, 12, 40, 17), (80, 16, 32, 8, 5, 6, 3)])
def test_source_encoder_only_train_step_matches_oracle(V, E, H, S, T, B, N):
    """source-encoder-only (BUILDER-DEFINED training, as for the CNN mode): LSTM source encoder + rows of the free
    target matrix; one step vs oracle._source_only_gradients."""
    import sse_amd
    params = model_params("source-encoder-only", V, E, H, H, S, T, N=N, lr=0.9)
    m, p = make_pair(params, seed=2)
    st = O.new_optimizer_state(p)
    rng = np.random.RandomState(7)
    src = np.repeat(random_ids(rng, B // 2, T, V, 0.5), 2, axis=0)
    rows = rng.randint(0, N, size=B).astype(np.int32)
    z = np.tile(np.array([1.0, 0.0], np.float32), B // 2)
    want = O.train_step(p, st, params, src, rows, z, 0.9)
    got = m.train_step(src, rows, z)
    assert got[0] == pytest.approx(float(want[0]), rel=LOSS_REL, abs=1e-6) and got[1] == pytest.approx(float(want[1]), abs=1e-6)
    vars_ = m.get_variables(with_slots=True)
    for name, w in p.items():
        assert np.abs(vars_[name].reshape(w.shape) - w).max() < 2e-4, name
        assert np.abs(vars_[name + "/Adagrad"].reshape(w.shape) - st[name]).max() < 2e-4, name
    with pytest.raises(ValueError):
        m.train_step(src, np.zeros_like(src), z)
    bad = rows.copy()
    bad[0] = N
    with pytest.raises(sse_amd.SSEError):
        m.train_step(src, bad, z)


@pytest.mark.parametrize("mode", ["dual-encoder", "source_only_cnn"])
def test_train_step_by_rows_equals_train_step_by_ids(mode):
    """sse_corpus_upload + sse_train_step_rows (batches as row numbers, ids gathered on the device) == sse_train_step on
    the gathered id matrices: the first loss bit for bit, afterwards to the run-to-run noise of the step itself (the
    embedding gradient is accumulated with float atomics, so two runs of the SAME step differ in the last bits);
    a row outside the corpus is an error."""
    import sse_amd
    V, T, N = 150, 16, 23
    params = model_params(mode, V, 24, 64, 64, 32, T, N=N, lr=0.5)
    rng = np.random.RandomState(3)
    src_corpus = random_ids(rng, 40, T, V, 0.5)
    tgt_corpus = random_ids(rng, N, T, V, 0.5)
    table = mode == "source_only_cnn"
    ma, p = make_pair(params, seed=2)
    mb, _ = make_pair(params, seed=2)
    mb.handle.corpus_upload(0, src_corpus)
    if not table:
        mb.handle.corpus_upload(1, tgt_corpus)
    for step in range(3):
        sr = np.repeat(rng.randint(0, 40, size=6), 2).astype(np.int32)
        tr = rng.randint(0, N, size=12).astype(np.int32)
        z = np.tile(np.array([1.0, 0.0], np.float32), 6)
        la = ma.train_step(src_corpus[sr], tr if table else tgt_corpus[tr], z)
        lb = mb.handle.train_step_rows(sr, tr, z)
        assert la == lb if step == 0 else la == pytest.approx(lb, rel=1e-5)
    ga, gb = ma.get_variables(with_slots=True), mb.get_variables(with_slots=True)
    for k in ga:
        assert np.abs(ga[k] - gb[k]).max() < 1e-5, k
    with pytest.raises(sse_amd.SSEError):
        mb.handle.train_step_rows(np.array([0, 40], np.int32), np.array([0, 1], np.int32), np.array([1, 0], np.float32))


def test_split_operand_training_converges_like_the_float32_step():
    """the opt-in split-operand train step (default since round 4: fp32).  300 steps of the same seeded
    batches on both arithmetics: the loss curves stay together (the split path is a ~4e-6-per-product perturbation, not
    a different optimiser), both fall, and the final weights agree to the level two float32 runs with different summation
    orders would."""
    params = model_params("dual-encoder", 400, 50, 96, 96, 64, 12, lr=0.3)
    (mx, _), (mf, _) = make_pair(params, seed=31), make_pair(params, seed=31)
    split_bf16_training(mx)
    exact_fp32_training(mf)
    rng = np.random.RandomState(3)
    corpus_s = random_ids(rng, 256, 12, 400, 0.4)
    corpus_t = random_ids(rng, 256, 12, 400, 0.4)
    lx, lf = [], []
    for step in range(300):
        rows = rng.randint(0, 256, size=64)
        neg = rng.randint(0, 256, size=64)
        src = np.repeat(corpus_s[rows], 2, axis=0)
        tgt = np.empty((128, 12), np.int32)
        tgt[0::2], tgt[1::2] = corpus_t[rows], corpus_t[neg]
        z = np.tile(np.array([1.0, 0.0], np.float32), 64)
        lx.append(mx.train_step(src, tgt, z)[0])
        lf.append(mf.train_step(src, tgt, z)[0])
    lx, lf = np.array(lx), np.array(lf)
    assert lf[-20:].mean() < 0.5 * lf[:20].mean() and lx[-20:].mean() < 0.5 * lx[:20].mean()      # both learn
    assert np.abs(lx[:50] - lf[:50]).max() < 2e-3 * max(1.0, lf[:50].max())                       # same trajectory early on
    assert abs(lx[-50:].mean() - lf[-50:].mean()) < 0.05 * max(lf[-50:].mean(), 0.05)             # same place at the end
    vx, vf = mx.get_variables(), mf.get_variables()
    for k in vx:
        assert np.abs(vx[k] - vf[k]).max() < 5e-2 * max(1.0, np.abs(vf[k]).max()), k


def test_restore_prefers_the_newer_of_npz_and_tf_index(tmp_path):
    """a stale converted .npz must not shadow a TensorFlow checkpoint re-written under the same prefix."""
    import os
    import time
    from sse_amd import tf_checkpoint
    params = model_params("shared-encoder", 60, 8, 16, 16, 8, 5)
    m, p = make_pair(params, seed=3)
    prefix = str(tmp_path / "SSE-LSTM.ckpt-5")
    m.handle.global_step = 5
    m.saver.save(None, str(tmp_path / "SSE-LSTM.ckpt"), global_step=5)            # writes <prefix>.npz (old weights)
    newer = {k: (v + 1.0).astype(np.float32) for k, v in m.get_variables(with_slots=True).items()}
    newer["global_step"] = np.array(9, np.int64)
    newer["learning_rate"] = np.array(0.25, np.float32)
    time.sleep(0.05)
    tf_checkpoint.write_bundle(prefix, newer)
    os.utime(prefix + ".index", (time.time() + 5, time.time() + 5))
    m.saver.restore(None, prefix)
    assert m.saver.restored_from.endswith(".index") and m.handle.global_step == 9
    assert np.array_equal(m.get_variables()["word_embedding"], newer["word_embedding"])
    os.utime(prefix + ".npz", (time.time() + 10, time.time() + 10))             # now the .npz is the newer one
    m.saver.restore(None, prefix)
    assert m.saver.restored_from.endswith(".npz") and m.handle.global_step == 5


def test_tape_size_limit_is_an_explicit_error():
    """a batch whose gate tape [T][rows/32][Hp/32][5][1024] floats reaches 2 GiB (32-bit offsets in the BPTT
    kernels) must be refused with a message that names the limit, not with a bare hipErrorInvalidValue."""
    import sse_amd
    T, B = 32, 16384                                                      # 32 * 512 * 8 * 5 * 1024 * 4 B = 2.5 GiB per encoder
    params = model_params("dual-encoder", 400, 50, 256, 256, 256, T)
    m, _ = make_pair(params, seed=1)
    rng = np.random.RandomState(0)
    src, tgt, z = _batch(rng, B, T, 400, pad_frac=0.0)
    with pytest.raises(sse_amd.SSEError) as e:
        m.train_step(src, tgt, z)
    assert "too large" in str(e.value) and "2 GiB" in str(e.value)
    src, tgt, z = _batch(rng, 256, T, 400)                               # the handle stays usable
    loss, _ = m.train_step(src, tgt, z)
    assert np.isfinite(loss)

"""Second, independent implementation of the WHOLE train step (sse_model.py:150-302,355-364) -- torch autograd on
CPU + torch.optim.Adagrad -- against which the numpy oracle's post-step variables and Adagrad slots are checked.

TensorFlow 1.x cannot run here, so this is the strongest pin the model half of the oracle can get: nothing below
shares code with oracle/sse_oracle.py (the gradients come from autograd, the update from torch's optimiser, the
sparse embedding update from torch's sparse Adagrad path which coalesces duplicate ids like TF's
`sparse_apply_adagrad` after `_deduplicate_indexed_slices`).  TF-1.x semantics restated here, each from the op's
documentation: BasicLSTMCell gate order i,j,f,o with forget_bias 1.0 added at run time; `tf.nn.l2_normalize` =
x * rsqrt(max(sum x^2, 1e-12)); `weighted_cross_entropy_with_logits(pos_weight=1)`; `clip_by_global_norm` with the
norm over the raw (concatenated, un-deduplicated) IndexedSlices values of BOTH embedding lookups;
AdagradOptimizer(initial_accumulator_value=0.1)."""
import numpy as np
import pytest
import torch

from oracle import sse_oracle as O


def _lstm_last(x, K, b, H):
    """static_rnn(BasicLSTMCell) in torch ops, TF layout: g = [x_t, h] @ K + b, split i,j,f,o (sse_model.py:240-242)."""
    B, T, _ = x.shape
    h = torch.zeros(B, H, dtype=x.dtype)
    c = torch.zeros(B, H, dtype=x.dtype)
    for t in range(T):
        g = torch.cat([x[:, t], h], dim=1) @ K + b
        i, j, f, o = torch.split(g, H, dim=1)
        c = c * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * torch.tanh(j)
        h = torch.tanh(c) * torch.sigmoid(o)
    return h


def _tf_l2_normalize(x):
    return x * torch.rsqrt(torch.clamp((x * x).sum(dim=-1, keepdim=True), min=1e-12))


def _torch_train_step(p0, cfg, src_ids, tgt_ids, labels, lr, dtype=torch.float32, clip_norm=5.0):
    mode = cfg["network_mode"]
    V = int(cfg["vocab_size"])
    var = {k: torch.tensor(v, dtype=dtype, requires_grad=True) for k, v in p0.items()}
    emb = var["word_embedding"]
    z = torch.tensor(labels, dtype=dtype)
    enc, looked = {}, {}
    for side, ids in (("src", src_ids), ("tgt", tgt_ids)):
        scope = O.lstm_scope(mode, side)            # names only (the variable-scope map of sse_model.py:220-272)
        K = var[scope + "/rnn/basic_lstm_cell/kernel"]
        b = var[scope + "/rnn/basic_lstm_cell/bias"]
        x = emb[torch.from_numpy(ids.astype(np.int64))]      # tf.nn.embedding_lookup, sse_model.py:163-164
        x.retain_grad()                                       # = the IndexedSlices values of this lookup
        looked[side] = x
        h = _lstm_last(x, K, b, K.shape[1] // 4)
        enc[side] = _tf_l2_normalize(h @ var[O.proj_name(mode, side)])
    logits = 64.0 * (enc["src"] * enc["tgt"]).sum(dim=-1)                      # sse_model.py:290,298
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, z)    # pos_weight 1, reduce_mean
    s = torch.sigmoid(logits)
    acc = (z * torch.floor(s + 0.1)).mean() + ((1 - z) * torch.floor(1.1 - s)).mean()   # sse_model.py:302
    loss.backward()
    # clip_by_global_norm(grads, 5.0): IndexedSlices enter with their raw values (sse_model.py:362)
    sq = sum(float((v.grad.double() ** 2).sum()) for k, v in var.items() if k != "word_embedding")
    sq += sum(float((x.grad.double() ** 2).sum()) for x in looked.values())
    gnorm = np.sqrt(sq)
    scale = clip_norm * min(1.0 / gnorm, 1.0 / clip_norm)
    # AdagradOptimizer(lr).apply_gradients: dense vars densely, the embedding sparsely (sse_model.py:359-363)
    opt = torch.optim.Adagrad(list(var.values()), lr=float(lr), initial_accumulator_value=0.1, eps=0.0)
    for k, v in var.items():
        if k == "word_embedding":
            idx = torch.from_numpy(np.concatenate([src_ids.reshape(-1), tgt_ids.reshape(-1)]).astype(np.int64))
            rows = torch.cat([looked["src"].grad.reshape(-1, emb.shape[1]), looked["tgt"].grad.reshape(-1, emb.shape[1])])
            v.grad = torch.sparse_coo_tensor(idx[None, :], rows * scale, size=(V, emb.shape[1]))
        else:
            v.grad = v.grad * scale
    opt.step()
    new = {k: v.detach().numpy() for k, v in var.items()}
    slots = {k: opt.state[v]["sum"].numpy() for k, v in var.items()}
    return float(loss.detach()), float(acc.detach()), new, slots, gnorm


@pytest.mark.parametrize("mode", ["dual-encoder", "shared-encoder"])
@pytest.mark.parametrize("clip", [True, False])
def test_oracle_train_step_matches_torch_autograd(mode, clip):
    cfg = dict(vocab_size=61, embedding_size=10, encoding_size=8, src_cell_size=12, tgt_cell_size=12 if mode != "dual-encoder" else 9,
               network_mode=mode, targetSpaceSize=5)
    p = O.init_params(cfg, seed=11)
    rng = np.random.RandomState(4)
    for k in p:
        if k.endswith("/bias"):
            p[k] = rng.uniform(-0.2, 0.2, size=p[k].shape).astype(np.float32)
    B, T = 12, 6
    src = rng.randint(0, 61, size=(B, T)).astype(np.int32)
    tgt = rng.randint(0, 61, size=(B, T)).astype(np.int32)
    src[:, 0] = 0                                           # duplicate ids inside one lookup ...
    tgt[:, :2] = 0                                          # ... and across the two lookups (PAD on both sides)
    src[1::2] = src[0::2]                                   # pos/neg rows share the source sequence (data.py:95-115)
    z = np.array([1, 0] * (B // 2), np.float32)
    lr = 0.9
    p0 = {k: v.copy() for k, v in p.items()}
    st = O.new_optimizer_state(p)
    # 64*cos logits make the raw norm O(100): the un-clipped branch (scale = 1) is reached by raising the threshold
    clip_norm = 5.0 if clip else 1.0e4
    old = O.MAX_GRAD_NORM
    try:
        O.MAX_GRAD_NORM = np.float32(clip_norm)
        loss, acc = O.train_step(p, st, cfg, src, tgt, z, lr)
    finally:
        O.MAX_GRAD_NORM = old
    tl, ta, tp, ts, gnorm = _torch_train_step(p0, cfg, src, tgt, z, lr, clip_norm=clip_norm)
    assert (gnorm > clip_norm) == clip, gnorm
    assert abs(float(loss) - tl) < 2e-6 * max(1.0, abs(tl))
    assert abs(float(acc) - ta) < 1e-6
    # clipped (the reference's regime): 1e-6 absolute.  Un-clipped, the updates are O(1) per element (g/sqrt(acc) ~ 1,
    # lr 0.9) and float32 summation order shows at a few ulp of 1.0; the float64 test below closes that to 1e-10.
    tol = 1e-6 if clip else 4e-6
    for k in p:
        assert np.abs(p[k] - tp[k]).max() < tol, (k, np.abs(p[k] - tp[k]).max())
        assert np.abs(st[k] - ts[k]).max() < tol * max(1.0, np.abs(ts[k]).max()), (k, np.abs(st[k] - ts[k]).max())
    # the un-deduplicated norm matters: the coalesced one is different on this batch
    _, _, g = O.gradients(p0, cfg, src, tgt, z)
    dense = O.dense_embedding_grad(g["word_embedding"], 61)
    assert abs(np.sqrt(np.sum(dense.astype(np.float64) ** 2)) - np.sqrt(np.sum(g["word_embedding"][1].astype(np.float64) ** 2))) > 1e-4
    assert abs(float(O.global_norm(g)) - gnorm) < 1e-5 * gnorm


def test_oracle_train_three_steps_match_torch_float64():
    """Three consecutive steps in float64 on both sides: slots and variables stay together to 1e-10."""
    cfg = dict(vocab_size=40, embedding_size=6, encoding_size=5, src_cell_size=7, tgt_cell_size=7,
               network_mode="dual-encoder", targetSpaceSize=5)
    p = {k: v.astype(np.float64) for k, v in O.init_params(cfg, seed=5).items()}
    rng = np.random.RandomState(9)
    old = O.F32
    try:
        O.F32 = np.float64
        O.FORGET_BIAS, O.L2_EPS, O.LOGIT_SCALE = np.float64(1.0), np.float64(1e-12), np.float64(64.0)
        O.MAX_GRAD_NORM, O.ADAGRAD_INIT_ACC = np.float64(5.0), np.float64(0.1)
        st = O.new_optimizer_state(p)
        tp = {k: v.copy() for k, v in p.items()}
        # torch side keeps its own slots across steps by re-creating the optimiser state from the returned sums
        tslots = None
        for step in range(3):
            src = rng.randint(0, 40, size=(8, 5)).astype(np.int32)
            tgt = rng.randint(0, 40, size=(8, 5)).astype(np.int32)
            z = np.array([1, 0] * 4, np.float64)
            O.train_step(p, st, cfg, src, tgt, z, 0.5)
            var = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in tp.items()}
            _, _, new, slots, _ = _torch_step_with_slots(var, tslots, cfg, src, tgt, z, 0.5)
            tp, tslots = new, slots
            for k in p:
                assert np.abs(p[k] - tp[k]).max() < 1e-10, (step, k)
                assert np.abs(st[k] - tslots[k]).max() < 1e-10, (step, k)
    finally:
        O.F32 = old
        O.FORGET_BIAS, O.L2_EPS, O.LOGIT_SCALE = old(1.0), old(1e-12), old(64.0)
        O.MAX_GRAD_NORM, O.ADAGRAD_INIT_ACC = old(5.0), old(0.1)


def _torch_step_with_slots(var, slots_in, cfg, src_ids, tgt_ids, labels, lr):
    """As _torch_train_step, but carrying the Adagrad accumulators over from a previous step."""
    mode = cfg["network_mode"]
    V = int(cfg["vocab_size"])
    emb = var["word_embedding"]
    dtype = emb.dtype
    z = torch.tensor(labels, dtype=dtype)
    enc, looked = {}, {}
    for side, ids in (("src", src_ids), ("tgt", tgt_ids)):
        scope = O.lstm_scope(mode, side)
        K, b = var[scope + "/rnn/basic_lstm_cell/kernel"], var[scope + "/rnn/basic_lstm_cell/bias"]
        x = emb[torch.from_numpy(ids.astype(np.int64))]
        x.retain_grad()
        looked[side] = x
        enc[side] = _tf_l2_normalize(_lstm_last(x, K, b, K.shape[1] // 4) @ var[O.proj_name(mode, side)])
    logits = 64.0 * (enc["src"] * enc["tgt"]).sum(dim=-1)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, z)
    loss.backward()
    sq = sum(float((v.grad ** 2).sum()) for k, v in var.items() if k != "word_embedding")
    sq += sum(float((x.grad ** 2).sum()) for x in looked.values())
    gnorm = np.sqrt(sq)
    scale = 5.0 * min(1.0 / gnorm, 1.0 / 5.0)
    opt = torch.optim.Adagrad(list(var.values()), lr=float(lr), initial_accumulator_value=0.1, eps=0.0)
    if slots_in is not None:
        for k, v in var.items():
            opt.state[v]["sum"] = torch.tensor(slots_in[k], dtype=dtype)
    E = emb.shape[1]
    for k, v in var.items():
        if k == "word_embedding":
            idx = torch.from_numpy(np.concatenate([src_ids.reshape(-1), tgt_ids.reshape(-1)]).astype(np.int64))
            rows = torch.cat([looked["src"].grad.reshape(-1, E), looked["tgt"].grad.reshape(-1, E)])
            v.grad = torch.sparse_coo_tensor(idx[None, :], rows * scale, size=(V, E))
        else:
            v.grad = v.grad * scale
    opt.step()
    return float(loss.detach()), 0.0, {k: v.detach().numpy() for k, v in var.items()}, \
        {k: opt.state[v]["sum"].numpy() for k, v in var.items()}, gnorm

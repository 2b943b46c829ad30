"""Shared helpers for the GPU parity tests."""
import numpy as np

from oracle import sse_oracle as O


def model_params(mode="dual-encoder", V=500, E=50, Hs=256, Ht=256, S=256, T=32, N=7, lr=0.9):
    return dict(forward_only=False, network_mode=mode, predict_nbest=10, max_seq_length=T, vocab_size=V,
                embedding_size=E, encoding_size=S, src_cell_size=Hs, tgt_cell_size=Ht, learning_rate=lr,
                learning_rate_decay_factor=0.99, targetSpaceSize=N)


def make_pair(params, seed=0, bias_scale=0.2):
    """(sse_amd.SSEModel on the GPU, oracle parameter dict) holding identical weights."""
    import sse_amd
    p = O.init_params(params, seed=seed)
    rng = np.random.RandomState(seed + 100)
    for k in p:
        if k.endswith("/bias"):
            p[k] = rng.uniform(-bias_scale, bias_scale, size=p[k].shape).astype(np.float32)
    m = sse_amd.SSEModel(params)
    m.set_variables(p)
    return m, p


def random_ids(rng, B, T, V, pad_frac=0.0):
    ids = rng.randint(2, V, size=(B, T)).astype(np.int32)
    ids[:, -1] = 1
    if pad_frac > 0:
        for b in range(B):
            npad = rng.randint(0, max(1, int(T * pad_frac)) + 1)
            ids[b, :npad] = 0
    return ids

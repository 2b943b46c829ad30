#!/usr/bin/env python
"""Train-step timing at the reference's DEFAULT hyper-parameters (sse_train.py:60-74: E=50, H=96, S=64, T=80,
batch_size=64 -> 128 pair rows) and at a large batch of the same model."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sse_amd  # noqa: E402

V, E, H, S, T = 32000, 50, 96, 64, 80
params = dict(forward_only=False, network_mode="dual-encoder", predict_nbest=10, max_seq_length=T, vocab_size=V,
              embedding_size=E, encoding_size=S, src_cell_size=H, tgt_cell_size=H, learning_rate=0.9,
              learning_rate_decay_factor=0.99, targetSpaceSize=571)
m = sse_amd.SSEModel(params)
m.init_variables(seed=0)
rng = np.random.RandomState(0)
for B in [int(x) for x in (sys.argv[1:] or ["128", "8192"])]:
    src = np.repeat(rng.randint(2, V, size=(B // 2, T)).astype(np.int32), 2, axis=0)
    tgt = rng.randint(2, V, size=(B, T)).astype(np.int32)
    z = np.tile(np.array([1.0, 0.0], np.float32), B // 2)
    for _ in range(3):
        m.train_step(src, tgt, z)
    n = 20 if B <= 1024 else 5
    t0 = time.perf_counter()
    for _ in range(n):
        loss, acc = m.train_step(src, tgt, z)
    dt = (time.perf_counter() - t0) / n
    print("reference defaults (H=96,T=80) B_rows=%d: %.3f ms/step  %.0f steps/s  %.0f pair-rows/s" % (B, dt * 1e3, 1 / dt, B / dt))

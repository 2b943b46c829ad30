#!/usr/bin/env python
"""What the any-shape LSTM path (csrc/lstm_generic.hip) costs: train step and encode at shapes the fused kernels do not take,
and -- option train_generic -- at the reference's default shape next to the fused step."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sse_amd  # noqa: E402


def model(E, H, S, T, V=32000):
    p = dict(forward_only=False, network_mode="dual-encoder", predict_nbest=10, max_seq_length=T, vocab_size=V, embedding_size=E,
             encoding_size=S, src_cell_size=H, tgt_cell_size=H, learning_rate=0.9, learning_rate_decay_factor=0.99, targetSpaceSize=571)
    m = sse_amd.SSEModel(p)
    m.init_variables(seed=0)
    return m


def batch(B, T, V=32000):
    rng = np.random.RandomState(0)
    src = np.repeat(rng.randint(2, V, size=(B // 2, T)).astype(np.int32), 2, axis=0)
    tgt = rng.randint(2, V, size=(B, T)).astype(np.int32)
    return src, tgt, np.tile(np.array([1.0, 0.0], np.float32), B // 2)


def time_step(m, B, T, n=5):
    src, tgt, z = batch(B, T)
    for _ in range(2):
        m.train_step(src, tgt, z)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        m.train_step(src, tgt, z)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, E, H, S, T, B, force in (("reference default shape, fused kernels", 50, 96, 64, 80, 128, 0),
                                   ("reference default shape, any-shape path (option train_generic)", 50, 96, 64, 80, 128, 1),
                                   ("--src_cell_size=300 --tgt_cell_size=300", 50, 300, 64, 80, 128, 0),
                                   ("--embedding_size=100", 100, 96, 64, 80, 128, 0),
                                   ("cell size 512, 1024 pair rows, T = 32", 50, 512, 256, 32, 1024, 0)):
    m = model(E, H, S, T)
    m.handle.set_option("train_generic", force)
    print("train step  E=%-3d H=%-3d S=%-3d T=%-2d rows=%-4d  %-66s %8.2f ms" % (E, H, S, T, B, name, time_step(m, B, T)))
    m.handle.close()
for name, E, H, S, T, B in (("cell size 700 (fused inference stops at 512)", 50, 700, 256, 32, 4096), ("encoding_size 1024", 50, 256, 1024, 32, 4096)):
    m = model(E, H, S, T)
    ids = batch(B, T)[1]
    m.encode_source(ids[:64])
    t0 = time.perf_counter()
    m.encode_source(ids)
    print("encode      E=%-3d H=%-3d S=%-4d T=%-2d rows=%-4d %-66s %8.2f ms (host buffers)" % (E, H, S, T, B, name, (time.perf_counter() - t0) * 1e3))
    m.handle.close()

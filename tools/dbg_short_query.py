#!/usr/bin/env python
"""Reproducer: single 8-token query, cluster LSTM kernel, 1.25 M-row index: kernel trace of the end-to-end call."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sse_amd  # noqa: E402

V, E, H, S, T, N = 32000, 50, 256, 256, 32, 1250000
params = dict(forward_only=True, network_mode="dual-encoder", predict_nbest=10, max_seq_length=T, vocab_size=V,
              embedding_size=E, encoding_size=S, src_cell_size=H, tgt_cell_size=H, learning_rate=0.9,
              learning_rate_decay_factor=0.99, targetSpaceSize=571)
m = sse_amd.SSEModel(params)
m.init_variables(seed=0)
hh = m.handle
rng = np.random.RandomState(0)
dense = rng.randint(2, V, size=(1, T)).astype(np.int32)
dense[:, -1] = 1
short = np.zeros((1, T), np.int32)
short[0, -9:] = dense[0, -9:]
t = torch.nn.functional.normalize(torch.randn((N, S), device="cuda:0"), dim=1)
hh.index_set_dev(t.data_ptr(), N, S)
for name, ids in (("dense", dense), ("short", short)):
    hh.encode_score_topk(0, ids, False, 10)
    t0 = time.perf_counter()
    for _ in range(10):
        hh.encode_score_topk(0, ids, False, 10)
    print(name, (time.perf_counter() - t0) / 10 * 1e3, "ms")

#!/usr/bin/env python
"""Timing of the MFMA cluster LSTM kernel (lstm_cluster.hip) at B rows, dense T=32, configs[1] model."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sse_amd
V, E, H, S, T = 32000, 50, 256, 256, 32
params = dict(forward_only=True, network_mode="dual-encoder", predict_nbest=10, max_seq_length=T, vocab_size=V,
              embedding_size=E, encoding_size=S, src_cell_size=H, tgt_cell_size=H, learning_rate=0.9,
              learning_rate_decay_factor=0.99, targetSpaceSize=571)
m = sse_amd.SSEModel(params)
m.init_variables(seed=0)
rng = np.random.RandomState(0)
if os.environ.get("WT"):
    m.handle.set_option("lstm_cluster_write_through", 1)
for B in [int(x) for x in (sys.argv[1:] or ["600"])]:
    ids = rng.randint(2, V, size=(B, T)).astype(np.int32)
    m.encode_source(ids)
    ts = []
    for _ in range(int(os.environ.get("N", "5"))):
        t0 = time.perf_counter(); m.encode_source(ids); ts.append(time.perf_counter() - t0)
    print("B=%d: %.3f ms (fallbacks %d)" % (B, sorted(ts)[len(ts) // 2] * 1e3, m.handle.get_counter("lstm_persist_fallbacks")))

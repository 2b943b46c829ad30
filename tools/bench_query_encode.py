#!/usr/bin/env python
"""Encode latency of 1 .. 32 queries (configs[1] model, dense T = 32): cluster kernel vs few-sequences kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import sse_amd  # noqa: E402

V, E, H, S, T = 32000, 50, 256, 256, 32
if len(sys.argv) > 1:
    H, S, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
params = dict(forward_only=True, network_mode="dual-encoder", predict_nbest=10, max_seq_length=T, vocab_size=V,
              embedding_size=E, encoding_size=S, src_cell_size=H, tgt_cell_size=H, learning_rate=0.9,
              learning_rate_decay_factor=0.99, targetSpaceSize=571)
m = sse_amd.SSEModel(params)
m.init_variables(seed=0)
h = m.handle
dev = torch.device("cuda:0")
for B in (1, 4, 8, 32):
    ids = torch.randint(2, V, (B, T), device=dev, dtype=torch.int32)
    out = torch.empty((B, S), device=dev)
    res = []
    for rows in (32, 0):
        h.set_option("lstm_persist_rows", rows)
        for _ in range(5):
            h.encode_dev(0, ids.data_ptr(), B, T, True, out.data_ptr())
        n = 50
        h.timer_record(0)
        for _ in range(n):
            h.encode_dev(0, ids.data_ptr(), B, T, True, out.data_ptr())
        h.timer_record(1)
        res.append(h.timer_elapsed_ms(0, 1) / n)
    print("H=%d T=%d B=%d: cluster kernel %.3f ms (%.1f us/step)   few-sequences kernel %.3f ms" % (H, T, B, res[0], res[0] / T * 1e3, res[1]))

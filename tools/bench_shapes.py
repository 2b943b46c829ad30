#!/usr/bin/env python
"""Exact fp32 encode at the reference's recipe shapes (every makefile recipe keeps the default cell size 96), 16384
device-resident rows: ms, seq/s, fraction of the fp32 MFMA peak over SURVEY 8d's algorithmic flops.
usage: tools/bench_shapes.py [E H S T]..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import sse_amd  # noqa: E402

V, B = 32000, 16384
shapes = [(50, 96, 64, 80), (50, 128, 64, 80), (40, 96, 50, 50), (30, 96, 64, 60), (40, 64, 50, 50), (50, 256, 256, 32)]
if len(sys.argv) > 4:
    a = [int(x) for x in sys.argv[1:]]
    shapes = [tuple(a[i:i + 4]) for i in range(0, len(a) - 3, 4)]
dev = torch.device("cuda:0")
for (E, H, S, T) in shapes:
    params = dict(forward_only=True, network_mode="dual-encoder", predict_nbest=10, max_seq_length=T, vocab_size=V,
                  embedding_size=E, encoding_size=S, src_cell_size=H, tgt_cell_size=H, learning_rate=0.9,
                  learning_rate_decay_factor=0.99, targetSpaceSize=571)
    m = sse_amd.SSEModel(params)
    m.init_variables(seed=0)
    h = m.handle
    ids = torch.randint(2, V, (B, T), device=dev, dtype=torch.int32)
    ids[:, -1] = 1
    out = torch.empty((B, S), device=dev)
    for _ in range(3):
        h.encode_dev(0, ids.data_ptr(), B, T, True, out.data_ptr())
    n = 10
    h.timer_record(0)
    for _ in range(n):
        h.encode_dev(0, ids.data_ptr(), B, T, True, out.data_ptr())
    h.timer_record(1)
    ms = h.timer_elapsed_ms(0, 1) / n
    flop = T * 8 * H * (E + H) + 2 * H * S
    print("E=%d H=%d S=%d T=%d B=%d: %.3f ms  %.0f seq/s  %.1f TFLOP/s algorithmic (%.3f of the fp32 MFMA peak)"
          % (E, H, S, T, B, ms, B / ms * 1e3, B * flop / ms / 1e9, B * flop / ms / 1e9 / 157.3))
    h.close()

#!/usr/bin/env python
"""Small driver for rocprofv3 passes: a few launches of the two dominant kernels
at bench sizes (LSTM encoder B=16384,T=32,H=S=256; scoring 8192 x 1.25M x 256)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import sse_amd  # noqa: E402

V, E, H, S, T = 32000, 50, 256, 256, 32
what = sys.argv[1] if len(sys.argv) > 1 else "both"
dev = torch.device("cuda:0")
params = dict(forward_only=True, network_mode="dual-encoder", predict_nbest=10, max_seq_length=T, vocab_size=V,
              embedding_size=E, encoding_size=S, src_cell_size=H, tgt_cell_size=H, learning_rate=0.9,
              learning_rate_decay_factor=0.99, targetSpaceSize=571)
m = sse_amd.SSEModel(params)
m.init_variables(seed=0)
h = m.handle
if what in ("both", "lstm"):
    B = int(os.environ.get("PROF_B", "16384"))
    ids = torch.randint(2, V, (B, T), device=dev, dtype=torch.int32)
    out = torch.empty((B, S), device=dev)
    for _ in range(4):
        h.encode_dev(0, ids.data_ptr(), B, T, True, out.data_ptr())
    torch.cuda.synchronize()
if what in ("both", "score"):
    N, Q = int(os.environ.get("PROF_N", "1250000")), int(os.environ.get("PROF_Q", "8192"))
    t = torch.nn.functional.normalize(torch.randn((N, S), device=dev), dim=1)
    q = torch.nn.functional.normalize(torch.randn((Q, S), device=dev), dim=1)
    h.index_set_dev(t.data_ptr(), N, S)
    os_ = torch.empty((Q, 10), dtype=torch.float64, device=dev)
    oi = torch.empty((Q, 10), dtype=torch.int64, device=dev)
    for _ in range(3):
        h.score_topk_dev(q.data_ptr(), Q, 10, os_.data_ptr(), oi.data_ptr())
    torch.cuda.synchronize()
h.synchronize()
print("done")

#!/usr/bin/env python
"""Single-query (demo / web, sse_demo.py:121-134) scoring against a large resident index: HBM-bound
regime, algorithmic bytes = N*S*4 per pass."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import sse_amd  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
S = 256
dev = torch.device("cuda:0")
params = dict(forward_only=True, network_mode="dual-encoder", predict_nbest=10, max_seq_length=8, vocab_size=50,
              embedding_size=8, encoding_size=S, src_cell_size=16, tgt_cell_size=16, learning_rate=0.9,
              learning_rate_decay_factor=0.99, targetSpaceSize=5)
h = sse_amd.SSEModel(params).handle
t = torch.empty((N, S), device=dev)
for i in range(0, N, 1_000_000):
    t[i:i + 1_000_000] = torch.nn.functional.normalize(torch.randn((min(1_000_000, N - i), S), device=dev), dim=1)
h.index_set_dev(t.data_ptr(), N, S)
del t
h.set_option("score_bf16", 0)                      # first the fp32-candidate sweep (4 bytes per index element)
for Q in (1, 8, 32):
    q = torch.nn.functional.normalize(torch.randn((Q, S), device=dev), dim=1)
    os_ = torch.empty((Q, 10), dtype=torch.float64, device=dev)
    oi = torch.empty((Q, 10), dtype=torch.int64, device=dev)
    h.score_topk_dev(q.data_ptr(), Q, 10, os_.data_ptr(), oi.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        h.score_topk_dev(q.data_ptr(), Q, 10, os_.data_ptr(), oi.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("Q=%d N=%d S=%d: %.3f ms/pass, %.2f TB/s of index streamed, %.3g scores/s"
          % (Q, N, S, dt * 1e3, N * S * 4 / dt / 1e12, Q * N / dt))

# the library default: bf16 candidate pass (option score_bf16 = 1): half the index bytes to stream, exact results
h.set_option("score_bf16", 1)
for Q in (1, 32):
    q = torch.nn.functional.normalize(torch.randn((Q, S), device=dev), dim=1)
    os_ = torch.empty((Q, 10), dtype=torch.float64, device=dev)
    oi = torch.empty((Q, 10), dtype=torch.int64, device=dev)
    ref_s, ref_i = torch.empty_like(os_), torch.empty_like(oi)
    h.set_option("score_bf16", 0)
    h.score_topk_dev(q.data_ptr(), Q, 10, ref_s.data_ptr(), ref_i.data_ptr())
    h.set_option("score_bf16", 1)
    h.score_topk_dev(q.data_ptr(), Q, 10, os_.data_ptr(), oi.data_ptr())
    torch.cuda.synchronize()
    same = bool(torch.equal(oi, ref_i) and torch.equal(os_, ref_s))
    t0 = time.perf_counter()
    for _ in range(5):
        h.score_topk_dev(q.data_ptr(), Q, 10, os_.data_ptr(), oi.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("Q=%d bf16 candidates: %.3f ms/pass, %.2f TB/s of bf16 index streamed, %.3g scores/s, identical to fp32 candidates: %s"
          % (Q, dt * 1e3, N * S * 2 / dt / 1e12, Q * N / dt, same))

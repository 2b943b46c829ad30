#!/usr/bin/env python
"""Scoring timing at ranking scale (BASELINE configs[3] shard: Q x 1.25 M x S): bf16 candidates (default) against fp32
candidates; results must be bit-identical.  usage: bench_score.py [Q] [N] [S]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sse_amd  # noqa: E402

Q = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1250000
S = int(sys.argv[3]) if len(sys.argv) > 3 else 256
params = dict(forward_only=True, network_mode="dual-encoder", predict_nbest=10, max_seq_length=4, vocab_size=50,
              embedding_size=8, encoding_size=S, src_cell_size=16, tgt_cell_size=16, learning_rate=0.9,
              learning_rate_decay_factor=0.99, targetSpaceSize=5)
h = sse_amd.SSEModel(params).handle
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
t = torch.nn.functional.normalize(torch.randn((N, S), generator=g, device=dev), dim=1)
q = torch.nn.functional.normalize(torch.randn((Q, S), generator=g, device=dev), dim=1)
h.index_set_dev(t.data_ptr(), N, S)
out = {}
for name, opts in (("bf16", dict(score_bf16=1, score_two_pass_rows=0)),
                   ("bf16 two-pass", dict(score_bf16=1, score_two_pass_rows=2147483647, score_two_pass_min_rows=0)),
                   ("fp32", dict(score_bf16=0, score_two_pass_rows=0))):
    for k_, v in opts.items():
        h.set_option(k_, v)
    s = torch.empty((Q, 10), dtype=torch.float64, device=dev)
    i = torch.empty((Q, 10), dtype=torch.int64, device=dev)
    h.score_topk_dev(q.data_ptr(), Q, 10, s.data_ptr(), i.data_ptr())
    torch.cuda.synchronize()
    n = int(os.environ.get("SSE_BENCH_PASSES", 5)) if name != "fp32" else 2
    t0 = time.perf_counter()
    for _ in range(n):
        h.score_topk_dev(q.data_ptr(), Q, 10, s.data_ptr(), i.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    out[name] = (s.clone(), i.clone())
    peak = 2500.0 if name != "fp32" else 157.3
    print("%-16s Q=%d N=%d S=%d: %.3f ms/pass, %.3g scores/s, %.0f TFLOP/s algorithmic = %.2f of the %s peak"
          % (name, Q, N, S, dt * 1e3, Q * N / dt, 2.0 * S * Q * N / dt / 1e12, 2.0 * S * Q * N / dt / 1e12 / peak,
             "bf16" if name != "fp32" else "fp32"))
ref = out["fp32"]
print("bf16 identical to fp32 candidates:", bool(torch.equal(out["bf16"][0], ref[0]) and torch.equal(out["bf16"][1], ref[1])),
      "| two-pass identical:", bool(torch.equal(out["bf16 two-pass"][0], ref[0]) and torch.equal(out["bf16 two-pass"][1], ref[1])),
      "| two-pass calls", h.get_counter("score_two_pass_calls"))
print("second chance / collect / brute force:", h.get_counter("score_bf16_second_chance_queries"), h.get_counter("score_collect_queries"),
      h.get_counter("score_bruteforce_queries"))

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

# ---- end to end, one query: token ids in -> top-10 out (sse_demo.py:112-134), encoder + scorer in one call
import numpy as np  # noqa: E402
V, E, H, T = 32000, 50, 256, 32
params = dict(forward_only=True, network_mode="dual-encoder", predict_nbest=10, max_seq_length=T, vocab_size=V,
              embedding_size=E, encoding_size=S, src_cell_size=H, tgt_cell_size=H, learning_rate=0.9,
              learning_rate_decay_factor=0.99, targetSpaceSize=571)
m = sse_amd.SSEModel(params)
m.init_variables(seed=0)
hh = m.handle
if os.environ.get("WT"):                                    # the any-placement (write-through) publish path of the cluster kernels
    hh.set_option("lstm_cluster_write_through", 1)
rng = np.random.RandomState(0)
dense = rng.randint(2, V, size=(1, T)).astype(np.int32)
dense[:, -1] = 1
short = np.zeros((1, T), np.int32)
short[0, -9:] = dense[0, -9:]                               # 8 tokens + EOS, left-padded as sse_demo.py:116-119 does
for n_idx in (571, N):
    t = torch.nn.functional.normalize(torch.randn((n_idx, S), device=dev), dim=1)
    hh.index_set_dev(t.data_ptr(), n_idx, S)
    del t
    for name, ids in (("dense T=32", dense), ("8 tokens", short)):
        for kern, persist, small in (("cluster (weights in LDS)", 32, 1024), ("few-sequences", 0, 1024), ("matrix", 0, 0)):
            hh.set_option("lstm_persist_rows", persist)
            hh.set_option("lstm_small_rows", small)
            hh.encode_score_topk(0, ids, False, 10)

            def med(fn, n=30):
                # median of per-call times: the HIP runtime stalls ONE call in a few thousand for tens of milliseconds
                # (a 35 ms gap with an idle GPU in the kernel trace, profiles/r03_notes.txt), which a mean over 20 calls
                # reports as +1.7 ms on every call
                ts = []
                for _ in range(n):
                    t0 = time.perf_counter()
                    fn()
                    ts.append(time.perf_counter() - t0)
                return sorted(ts)[n // 2]
            dt = med(lambda: hh.encode_score_topk(0, ids, False, 10))
            de = med(lambda: hh.encode(0, ids, False))
            cnt = {c: hh.get_counter(c) for c in ("lstm_persist_fallbacks", "score_bf16_second_chance_queries",
                                                  "score_collect_queries", "score_bruteforce_queries")}
            print("Q=1 end to end (%s, N=%d, %s LSTM kernel): %.3f ms per query (encode alone %.3f ms, host buffers both ways)  "
                  "[cumulative: cluster fallbacks %d, bf16 second chance %d, collect %d, brute force %d]"
                  % (name, n_idx, kern, dt * 1e3, de * 1e3, cnt["lstm_persist_fallbacks"], cnt["score_bf16_second_chance_queries"],
                     cnt["score_collect_queries"], cnt["score_bruteforce_queries"]))
# batch-size sweep of the encoder alone, the three kernels
for B in (1, 4, 32, 64, 128, 256, 512, 600, 1000, 1024):
    ids = rng.randint(2, V, size=(B, T)).astype(np.int32)
    ids[:, -1] = 1
    line = "encode B=%d dense T=32:" % B
    for kern, persist, small in (("cluster", 32, 4096), ("few-seq", 0, 4096), ("matrix", 0, 0)):
        hh.set_option("lstm_persist_rows", persist)
        hh.set_option("lstm_cluster_rows", 1024 if kern == "cluster" else 0)
        hh.set_option("lstm_small_rows", small)
        hh.encode(0, ids, True)
        t0 = time.perf_counter()
        for _ in range(10):
            hh.encode(0, ids, True)
        line += "  %s %.3f ms" % (kern, (time.perf_counter() - t0) / 10 * 1e3)
    print(line)

#!/usr/bin/env python
"""One query, token ids in -> top-10 out (sse_demo.py:112-134, webserver.py:124-161): the parts of the call, each the median
of N calls.  python tools/bench_latency.py [index rows] [calls]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import sse_amd  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
CALLS = int(sys.argv[2]) if len(sys.argv) > 2 else 51
S, V, E, H, T = 256, 32000, 64, 256, 32
dev = torch.device("cuda:0")
params = dict(forward_only=True, network_mode="dual-encoder", predict_nbest=10, max_seq_length=T, vocab_size=V,
              embedding_size=E, encoding_size=S, src_cell_size=H, tgt_cell_size=H, learning_rate=0.9,
              learning_rate_decay_factor=0.99, targetSpaceSize=571)
m = sse_amd.SSEModel(params)
m.init_variables(seed=0)
h = m.handle


def med(fn, n=CALLS):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[n // 2] * 1e3


rng = np.random.RandomState(0)
ids = rng.randint(2, V, size=(1, T)).astype(np.int32)
ids[:, -1] = 1
small = torch.nn.functional.normalize(torch.randn((571, S), device=dev), dim=1)
big = torch.nn.functional.normalize(torch.randn((N, S), device=dev), dim=1)
q = torch.nn.functional.normalize(torch.randn((1, S), device=dev), dim=1)
qh = q.cpu().numpy()
os1 = torch.empty((1, 10), dtype=torch.float64, device=dev)
oi1 = torch.empty((1, 10), dtype=torch.int64, device=dev)


def sweep_dev():
    h.score_topk_dev(q.data_ptr(), 1, 10, os1.data_ptr(), oi1.data_ptr())
    torch.cuda.synchronize()


for coop in (1, 0):
    h.set_option("lstm_cluster_coop", coop)
    print("cluster kernels launched %s" % ("cooperatively" if coop else "plainly"))
    print("  encode (ids -> encoding, host buffers)             %.3f ms" % med(lambda: h.encode(0, ids, False)))
    for name, idx, n in (("571", small, 571), (str(N), big, N)):
        h.index_set_dev(idx.data_ptr(), n, S)
        print("  index of %s rows:" % name)
        print("    score, device buffers + synchronize              %.3f ms" % med(sweep_dev))
        print("    score, host buffers (sse_score_topk)             %.3f ms" % med(lambda: h.score_topk(qh, 10)))
        print("    ids -> top-10 (sse_encode_score_topk)            %.3f ms" % med(lambda: h.encode_score_topk(0, ids, False, 10)))
print("counters:", {c: h.get_counter(c) for c in ("lstm_persist_fallbacks", "score_bf16_second_chance_queries",
                                                  "score_collect_queries", "score_bruteforce_queries")})

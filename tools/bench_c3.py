#!/usr/bin/env python
"""configs[2] scoring on the real crosslingual rows (random-init weights): where the 1.5 ms of the ranking pass go --
bf16 candidates vs fp32 candidates, the second-chance / collect counters, margins of the top scores."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sse_amd  # noqa: E402

z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "crosslingual_full_ids.npz"))
src, tgt = z["src_ids"].astype(np.int32), z["tgt_ids"].astype(np.int32)
params = dict(forward_only=True, network_mode="dual-encoder", predict_nbest=10, max_seq_length=50, vocab_size=int(z["vocab_size"]),
              embedding_size=50, encoding_size=256, src_cell_size=256, tgt_cell_size=256, learning_rate=0.9,
              learning_rate_decay_factor=0.99, targetSpaceSize=len(tgt))
m = sse_amd.SSEModel(params)
m.init_variables(seed=0)
h = m.handle
dev = torch.device("cuda:0")
te = torch.from_numpy(m.encode_target(tgt)).to(dev)
se = torch.from_numpy(m.encode_source(src)).to(dev)
Q, N = len(src), len(tgt)
s = torch.empty((Q, 10), dtype=torch.float64, device=dev)
i = torch.empty((Q, 10), dtype=torch.int64, device=dev)
names = ("score_bf16_second_chance_queries", "score_collect_queries", "score_bruteforce_queries")
ref_ids = None
for bf, tp in ((1, 262144), (1, 0), (0, 0)):
    h.set_option("score_bf16", bf)
    h.set_option("score_two_pass_rows", tp)
    h.set_option("score_two_pass_min_rows", 0)
    h.index_set_dev(te.data_ptr(), N, 256)
    c0 = [h.get_counter(c) for c in names]
    h.score_topk_dev(se.data_ptr(), Q, 10, s.data_ptr(), i.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        h.score_topk_dev(se.data_ptr(), Q, 10, s.data_ptr(), i.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    c1 = [h.get_counter(c) for c in names]
    same = True if ref_ids is None else bool(torch.equal(i, ref_ids))
    ref_ids = i.clone() if ref_ids is None else ref_ids
    print("score_bf16=%d two_pass_rows=%d: %.3f ms/pass; per pass second chance %d, collect %d, brute force %d queries of %d; ids equal to the first variant: %s"
          % (bf, tp, dt * 1e3, (c1[0] - c0[0]) // 6, (c1[1] - c0[1]) // 6, (c1[2] - c0[2]) // 6, Q, same))
print("top-1 score range %.6f .. %.6f; median top-1 - top-10 margin %.2e" % (float(s[:, 0].min()), float(s[:, 0].max()), float((s[:, 0] - s[:, 9]).median())))
# the same shape with well-spread unit vectors (what a trained model's encodings look like to the candidate pass)
g = torch.Generator(device=dev).manual_seed(1)
t2 = torch.nn.functional.normalize(torch.randn((N, 256), generator=g, device=dev), dim=1)
q2 = torch.nn.functional.normalize(torch.randn((Q, 256), generator=g, device=dev), dim=1)
h.set_option("score_bf16", 1)
for NN in (N, 65536, 131072, 262144, 524288):
    t2 = torch.nn.functional.normalize(torch.randn((NN, 256), generator=g, device=dev), dim=1)
    h.index_set_dev(t2.data_ptr(), NN, 256)
    line = "random unit vectors, %d x %d:" % (Q, NN)
    for tp in (2147483647, 0):
        h.set_option("score_two_pass_rows", tp)
        h.score_topk_dev(q2.data_ptr(), Q, 10, s.data_ptr(), i.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            h.score_topk_dev(q2.data_ptr(), Q, 10, s.data_ptr(), i.data_ptr())
        torch.cuda.synchronize()
        line += "  %s %.3f ms/pass" % ("two-pass" if tp else "list sweep", (time.perf_counter() - t0) / 5 * 1e3)
    print(line)

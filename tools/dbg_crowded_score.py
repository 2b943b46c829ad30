#!/usr/bin/env python
"""Debug aid: the crowded / tied index of tests/test_gpu_score.py::test_two_pass_path_for_mid_size_indexes_is_exact, per path:
which queries differ from the oracle, and how."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sse_amd  # noqa: E402
from oracle import sse_oracle as O  # noqa: E402

Q, N, S, k = [int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (2048, 32060, 256, 10))]
crowd_n = int(sys.argv[5]) if len(sys.argv) > 5 else 700
rng = np.random.RandomState(Q + N)


def unit(n, s):
    x = rng.standard_normal((n, s)).astype(np.float32)
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


t = unit(N, S).astype(np.float64)
q = unit(Q, S)
t[N - 1] = t[17]
t[N // 2] = t[17]
q[3] = t[17].astype(np.float32)
base = q[5].astype(np.float64)
base /= np.linalg.norm(base)
crowd = rng.choice(np.arange(100, N - 100), crowd_n, replace=False)
for j, r in enumerate(crowd):
    u = rng.standard_normal(S)
    u -= u.dot(base) * base
    u /= np.linalg.norm(u)
    c = 1.0 - 1e-7 * j
    t[r] = c * base + np.sqrt(max(0.0, 1.0 - c * c)) * u
params = dict(forward_only=True, network_mode="dual-encoder", predict_nbest=10, max_seq_length=4, vocab_size=50, embedding_size=8,
              encoding_size=8, src_cell_size=16, tgt_cell_size=16, learning_rate=0.9, learning_rate_decay_factor=0.99, targetSpaceSize=7)
m = sse_amd.SSEModel(params)
m.init_variables(seed=0)
h = m.handle
h.set_option("score_bf16", 1)
h.index_upload(t, id_base=1000)
wsc, wids = O.topk(O.scores_f64(q, t), k)
names = ("score_bf16_second_chance_queries", "score_collect_queries", "score_bruteforce_queries")
for tp in (0, 262144):
    h.set_option("score_two_pass_rows", tp)
    h.set_option("score_two_pass_min_rows", 0)
    n0 = h.get_counter("score_two_pass_calls")
    c0 = [h.get_counter(c) for c in names]
    sc, ids = h.score_topk(q, k)
    c1 = [h.get_counter(c) for c in names]
    bad = np.where((ids != wids + 1000).any(axis=1))[0]
    print("two-pass calls %d;" % (h.get_counter("score_two_pass_calls") - n0), end=" ")
    print("two_pass_rows=%d: %d queries differ %s; counters (second chance, collect, brute force) %s; max |score diff| %.2e"
          % (tp, len(bad), bad[:10].tolist(), [b - a for a, b in zip(c0, c1)], float(np.abs(sc - wsc).max())))
    for b in bad[:3]:
        print("  query %d: got %s" % (b, (ids[b] - 1000).tolist()))
        print("           want %s" % wids[b].tolist())
        print("           got scores  %s" % ["%.12f" % x for x in sc[b]])
        print("           want scores %s" % ["%.12f" % x for x in wsc[b]])

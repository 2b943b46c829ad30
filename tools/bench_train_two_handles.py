#!/usr/bin/env python
"""Does the device pack two INDEPENDENT half-size train steps better than one full-size step?  Two handles on two threads,
4096 pair rows each, against one handle at 8192 rows (configs[1] model).  (Measurement for a chunk-pipelined train step:
forward of one chunk beside the backward of the other; profiles/r04_notes.txt.)"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sse_amd  # noqa: E402

V, E, H, S, T = 32000, 50, 256, 256, 32
params = dict(forward_only=False, network_mode="dual-encoder", predict_nbest=10, max_seq_length=T, vocab_size=V,
              embedding_size=E, encoding_size=S, src_cell_size=H, tgt_cell_size=H, learning_rate=0.9,
              learning_rate_decay_factor=0.99, targetSpaceSize=571)


def batch(rng, B):
    src = np.repeat(rng.randint(2, V, size=(B // 2, T)).astype(np.int32), 2, axis=0)
    tgt = rng.randint(2, V, size=(B, T)).astype(np.int32)
    return src, tgt, np.tile(np.array([1.0, 0.0], np.float32), B // 2)


def run(models, B, n=6):
    rng = np.random.RandomState(0)
    data = [batch(rng, B) for _ in models]
    for m, d in zip(models, data):
        for _ in range(2):
            m.train_step(*d)
    def loop(m, d):
        for _ in range(n):
            m.train_step(*d)
    th = [threading.Thread(target=loop, args=(m, d)) for m, d in zip(models, data)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    return (time.perf_counter() - t0) / n


ms = [sse_amd.SSEModel(params) for _ in range(2)]
for m in ms:
    m.init_variables(seed=0)
one = run(ms[:1], 8192)
two = run(ms, 4096)
four = run(ms, 2048, n=12)
print("one handle, 8192 rows: %.3f ms/step; two handles x 4096 rows concurrently: %.3f ms per pair of steps (%.2fx); "
      "two handles x 2048 rows: %.3f ms per pair x 2 = %.3f ms per 8192 rows"
      % (one * 1e3, two * 1e3, one / two, four * 1e3, 2 * four * 1e3))

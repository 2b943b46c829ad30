#!/usr/bin/env python
"""Real-data encode throughput (BASELINE configs[2] shapes: dual-encoder H=S=256 E=50, T=50) on the
token-id rows of rawdata-crosslingual produced by the reference's own data_utils
(tests/golden/crosslingual_ids.npz), replicated to index scale; with and without the exact
left-PAD prefix skip."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sse_amd  # noqa: E402

z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "crosslingual_ids.npz"))
V = int(z["vocab_size"])
params = dict(forward_only=True, network_mode="dual-encoder", predict_nbest=10, max_seq_length=50, vocab_size=V,
              embedding_size=50, encoding_size=256, src_cell_size=256, tgt_cell_size=256, learning_rate=0.9,
              learning_rate_decay_factor=0.99, targetSpaceSize=571)
m = sse_amd.SSEModel(params)
m.init_variables(seed=0)
reps = 7
tgt = np.tile(z["tgt_ids"], (reps, 1))          # 34,076 rows ~ the 32,060-target crosslingual index
src = np.tile(z["src_ids"], (28, 1))            # 16,800 rows ~ the 16,491 eval queries
for name, ids, enc in (("targets", tgt, m.encode_target), ("queries", src, m.encode_source)):
    nonpad = float((ids != 0).sum(1).mean())
    ref = None
    for x3 in (0, 1):                                       # exact fp32 kernel / opt-in split-bf16 kernel
        m.handle.set_option("lstm_x3", x3)
        for skip in (0, 1):
            m.handle.set_option("pad_skip", skip)
            enc(ids[:2048])
            t0 = time.perf_counter()
            out = enc(ids)
            dt = time.perf_counter() - t0
            if ref is None:
                ref = out
            print("%s: %d rows x T=50 (mean non-pad %.1f) lstm_x3=%d pad_skip=%d: %.1f ms  %.0f seq/s (host buffers in/out), max |d| vs first %.2e"
                  % (name, len(ids), nonpad, x3, skip, dt * 1e3, len(ids) / dt, float(np.abs(out - ref).max())))
    m.handle.set_option("lstm_x3", 0)

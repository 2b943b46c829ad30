#!/usr/bin/env python
"""Training-step timing (sse_train_step through the C ABI, host buffers in -> loss/acc out),
configs[1] model (dual-encoder E=50 H=S=256 T=32 V=32000)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sse_amd  # noqa: E402

V, E, H, S, T = 32000, 50, 256, 256, 32
params = dict(forward_only=False, network_mode="dual-encoder", predict_nbest=10, max_seq_length=T, vocab_size=V,
              embedding_size=E, encoding_size=S, src_cell_size=H, tgt_cell_size=H, learning_rate=0.9,
              learning_rate_decay_factor=0.99, targetSpaceSize=571)
m = sse_amd.SSEModel(params)
m.init_variables(seed=0)
if os.environ.get("SSE_TRAIN_SERIAL"):                 # profiling aid: isolated kernel durations
    m.handle.set_option("train_serial", 1)
if os.environ.get("SSE_TRAIN_ROWS"):                   # 32 / 64 rows per workgroup in the training forward (0: automatic)
    m.handle.set_option("lstm_train_rows", int(os.environ["SSE_TRAIN_ROWS"]))
if os.environ.get("SSE_TRAIN_PAIR_DEDUP"):            # 0: run the source encoder on every row of a paired batch
    m.handle.set_option("train_pair_dedup", int(os.environ["SSE_TRAIN_PAIR_DEDUP"]))
if os.environ.get("SSE_TRAIN_DK_X3"):                  # 0: fp32 MFMA weight-gradient GEMM
    m.handle.set_option("train_dk_x3", int(os.environ["SSE_TRAIN_DK_X3"]))
if os.environ.get("SSE_TRAIN_FWD_X3"):
    m.handle.set_option("train_fwd_x3", int(os.environ["SSE_TRAIN_FWD_X3"]))
if os.environ.get("SSE_TRAIN_BWD_X3"):
    m.handle.set_option("train_bwd_x3", int(os.environ["SSE_TRAIN_BWD_X3"]))
rng = np.random.RandomState(0)
for B in [int(x) for x in (sys.argv[1:] or ["128", "1024", "8192"])]:
    if os.environ.get("SSE_TRAIN_UNPAIRED"):
        src = rng.randint(2, V, size=(B, T)).astype(np.int32)
    else:
        src = np.repeat(rng.randint(2, V, size=(B // 2, T)).astype(np.int32), 2, axis=0)   # data.py:95-115: pos,neg share a source
    tgt = rng.randint(2, V, size=(B, T)).astype(np.int32)
    if os.environ.get("SSE_TRAIN_REALISTIC"):          # left padding (half the positions on average) and EOS, as data.py pads
        for arr in (src, tgt):
            arr[:, -1] = 1
            lens = rng.randint(2, T, size=arr.shape[0])
            if arr is src:
                lens = np.repeat(lens[0::2], 2)[:arr.shape[0]]
            arr[np.arange(T)[None, :] < (T - lens)[:, None]] = 0
    z = np.tile(np.array([1.0, 0.0], np.float32), B // 2)
    for _ in range(3):
        m.train_step(src, tgt, z)
    n = 20 if B <= 1024 else 5
    t0 = time.perf_counter()
    for _ in range(n):
        loss, acc = m.train_step(src, tgt, z)
    dt = (time.perf_counter() - t0) / n
    flops = 3 * 2 * B * (T * 8 * H * (E + H) + 2 * H * S)       # ~3x forward, two encoders
    print("B_rows=%d: %.3f ms/step, %.0f pair-rows/s, %.1f TFLOP/s algorithmic, loss %.4f" % (B, dt * 1e3, B / dt, flops / dt / 1e12, loss))

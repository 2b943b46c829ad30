#!/usr/bin/env python
"""Text-CNN encoder forward throughput (BASELINE configs[4] shapes: source_only_cnn, T=64, S=512, E=50),
inputs resident in HBM.  Algorithmic work 2E*sum((T-fs+1)*fs*nf) + 2*576*S = 11.24 MFLOP/sequence."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import sse_amd  # noqa: E402

V, E, S, T = 32000, 50, 512, 64
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device("cuda:0")
params = dict(forward_only=True, network_mode="source_only_cnn", predict_nbest=10, max_seq_length=T, vocab_size=V,
              embedding_size=E, encoding_size=S, src_cell_size=96, tgt_cell_size=96, learning_rate=0.9,
              learning_rate_decay_factor=0.99, targetSpaceSize=571)
m = sse_amd.SSEModel(params)
m.init_variables(seed=0)
h = m.handle
ids = torch.randint(2, V, (B, T), device=dev, dtype=torch.int32)
out = torch.empty((B, S), device=dev)
flop = 2 * E * sum((T - fs + 1) * fs * nf for fs, nf in zip((2, 3, 4, 5), (256, 128, 128, 64))) + 2 * 576 * S
for _ in range(3):
    h.encode_dev(0, ids.data_ptr(), B, T, True, out.data_ptr())
n = 10
h.timer_record(0)
for _ in range(n):
    h.encode_dev(0, ids.data_ptr(), B, T, True, out.data_ptr())
h.timer_record(1)
ms = h.timer_elapsed_ms(0, 1) / n
print("CNN encode B=%d T=%d S=%d: %.3f ms  %.0f seq/s  %.1f TFLOP/s algorithmic (%.1f%% of fp32 MFMA peak)"
      % (B, T, S, ms, B / ms * 1e3, B * flop / ms / 1e9, B * flop / ms / 1e9 / 157.3 * 100))

h.set_option("cnn_bf16", 1)
for _ in range(3):
    h.encode_dev(0, ids.data_ptr(), B, T, True, out.data_ptr())
h.timer_record(0)
for _ in range(n):
    h.encode_dev(0, ids.data_ptr(), B, T, True, out.data_ptr())
h.timer_record(1)
ms16 = h.timer_elapsed_ms(0, 1) / n
print("CNN encode, bf16 storage / fp32 accumulate (option cnn_bf16): %.3f ms  %.0f seq/s  %.1f TFLOP/s algorithmic"
      % (ms16, B / ms16 * 1e3, B * flop / ms16 / 1e9))
h.set_option("cnn_bf16", 0)

# ---- training step (builder-defined CNN pair loss, configs[4]): host ids in, loss/acc out
import time  # noqa: E402
import numpy as np  # noqa: E402

rng = np.random.RandomState(0)
for bf16 in (0, 1):
    h.set_option("cnn_bf16", bf16)
    for Bt in (1024, 8192):
        src = np.repeat(rng.randint(2, V, size=(Bt // 2, T)).astype(np.int32), 2, axis=0)
        src[:, -1] = 1                                       # every real sequence ends in EOS (sse_index.py:79-85): a hot embedding row
        rows = rng.randint(0, 571, size=Bt).astype(np.int32)
        z = np.tile(np.array([1.0, 0.0], np.float32), Bt // 2)
        for _ in range(2):
            m.train_step(src, rows, z)
        n = 10
        t0 = time.perf_counter()
        for _ in range(n):
            loss, acc = m.train_step(src, rows, z)
        dt = (time.perf_counter() - t0) / n
        print("CNN train step%s B_rows=%d: %.3f ms/step  %.0f pair-rows/s  (forward conv share %.2f ms at the encode rate above), loss %.4f"
              % (" (cnn_bf16: bf16 operands, fp32 masters/accumulate)" if bf16 else "", Bt, dt * 1e3, Bt / dt,
                 Bt * (ms16 if bf16 else ms) / B, loss))

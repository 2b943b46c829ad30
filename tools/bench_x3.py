#!/usr/bin/env python
"""Encode throughput of the configs[1] model (16384 x T=32, E=50, H=S=256): exact fp32 MFMA kernel vs the opt-in
split-bf16 kernel (option lstm_x3), inputs resident in HBM; and the distance between the two."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import sse_amd  # noqa: E402

V, E, H, S, T = 32000, 50, 256, 256, 32
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
if len(sys.argv) > 4:
    H, S, T = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
params = dict(forward_only=True, network_mode="dual-encoder", predict_nbest=10, max_seq_length=T, vocab_size=V,
              embedding_size=E, encoding_size=S, src_cell_size=H, tgt_cell_size=H, learning_rate=0.9,
              learning_rate_decay_factor=0.99, targetSpaceSize=571)
m = sse_amd.SSEModel(params)
m.init_variables(seed=0)
h = m.handle
dev = torch.device("cuda:0")
ids = torch.randint(2, V, (B, T), device=dev, dtype=torch.int32)
if os.environ.get("SSE_X3_PAD"):                     # left padding: lengths uniform in [2, T], rows sorted by length as sse_encode does
    lens = torch.sort(torch.randint(2, T + 1, (B,), device=dev))[0]
    ids[torch.arange(T, device=dev)[None, :] < (T - lens)[:, None]] = 0
flop = T * 8 * H * (E + H) + 2 * H * S
outs = []
for x3 in (0, 1):
    h.set_option("lstm_x3", x3)
    out = torch.empty((B, S), device=dev)
    for _ in range(3):
        h.encode_dev(0, ids.data_ptr(), B, T, True, out.data_ptr())
    n = 20
    h.timer_record(0)
    for _ in range(n):
        h.encode_dev(0, ids.data_ptr(), B, T, True, out.data_ptr())
    h.timer_record(1)
    ms = h.timer_elapsed_ms(0, 1) / n
    outs.append(out)
    print("H=%d S=%d T=%d B=%d  %s: %.3f ms  %.0f seq/s  %.1f TFLOP/s algorithmic%s"
          % (H, S, T, B, "lstm_x3 (3 bf16 MFMAs on hi+lo operands)" if x3 else "fp32 MFMA (exact)", ms, B / ms * 1e3, B * flop / ms / 1e9,
             "  (matrix pipe: %.0f TFLOP/s of bf16 work = %.2f of the 2.5 PF peak)" % (3 * B * flop / ms / 1e9, 3 * B * flop / ms / 1e9 / 2500)
             if x3 else "  (%.3f of the 157.3 TF fp32 peak)" % (B * flop / ms / 1e9 / 157.3)))
torch.cuda.synchronize()
d = (outs[0] - outs[1]).abs().max().item()
cos = (outs[0].double() * outs[1].double()).sum(dim=1).min().item()
print("max |x3 - fp32| over %d x %d normalised components: %.3e; min cosine %.10f" % (B, S, d, cos))

#!/usr/bin/env python
"""HBM-bound helper kernels against the 8 TB/s roofline: row L2-normalise (sse_l2_normalize_dev; 2*rows*S*4 bytes)
and the index re-layout done once by sse_index_set_dev (pack into MFMA fragment order + norm bound; reads rows*S*4,
writes rows*S*4)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import sse_amd  # noqa: E402
from tests.util import model_params  # noqa: E402

dev = torch.device("cuda:0")
m = sse_amd.SSEModel(model_params("dual-encoder", 50, 8, 16, 16, 256, 4))
m.init_variables(seed=0)
h = m.handle
for rows, S in ((4_000_000, 256), (1_250_000, 256), (16_384, 256), (4_000_000, 64)):
    x = torch.randn((rows, S), device=dev)
    out = torch.empty_like(x)
    for _ in range(2):
        h.l2_normalize_dev(x.data_ptr(), out.data_ptr(), rows, S)
    n = 10
    h.timer_record(0)
    for _ in range(n):
        h.l2_normalize_dev(x.data_ptr(), out.data_ptr(), rows, S)
    h.timer_record(1)
    ms = h.timer_elapsed_ms(0, 1) / n
    gb = 2.0 * rows * S * 4 / 1e9
    print("l2_normalize %9d x %3d: %.3f ms  %.2f TB/s (%.0f%% of 8 TB/s)" % (rows, S, ms, gb / ms, gb / ms / 8 * 100))
    ok = torch.allclose(out[:1000], torch.nn.functional.normalize(x[:1000], dim=1), atol=1e-6)
    assert ok
    for _ in range(2):
        h.index_set_dev(out.data_ptr(), rows, S)
    h.timer_record(0)
    for _ in range(5):
        h.index_set_dev(out.data_ptr(), rows, S)
    h.timer_record(1)
    ms = h.timer_elapsed_ms(0, 1) / 5
    print("index_set_dev %8d x %3d: %.3f ms  %.2f TB/s of read+write (%.0f%% of 8 TB/s)" % (rows, S, ms, gb / ms, gb / ms / 8 * 100))
    del x, out

#!/usr/bin/env python
"""Randomised bit-identity sweep of the LARGE-batch inference paths added in round 6 (tools/fuzz_parity.py stops at 256 rows):
device-resident ids through sse_encode_dev with every combination of option lstm_gate_split (lstm_fwd_gs.hip vs
lstm_fwd_kernel<2,1,1>) and option pad_sort_dev (off / adaptive / always) -- every combination must give the same BITS in the
caller's row order -- and a sample of rows against the CPU oracle.  Shapes: cell sizes 5 .. 256 (the gate-split kernel takes
H <= 128), E 3 .. 64, S 2 .. 300, T 3 .. 90, 64-row-tile batches (8193 .. 20000 rows) and mid batches, dense / left-padded /
mixed rows, rows that are all PAD, PADs in the middle of a row.   python tools/fuzz_large_batches.py [n_cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sse_amd  # noqa: E402
from oracle import sse_oracle as O  # noqa: E402
from tests.util import make_pair, model_params, random_ids  # noqa: E402


def main(n_cases=12, seed=0, verbose=True):
    import torch
    rng = np.random.RandomState(seed)
    dev = torch.device("cuda", 0)
    failures, worst = 0, 0.0
    t_start = time.time()
    for case in range(n_cases):
        mode = str(rng.choice(["dual-encoder", "shared-encoder"]))
        V = int(rng.choice([90, 500, 3000]))
        E = int(rng.choice([3, 8, 30, 40, 50, 63]))
        Hs = int(rng.choice([5, 17, 32, 40, 64, 65, 96, 100, 128, 200, 256]))
        Ht = Hs if mode == "shared-encoder" else int(rng.choice([7, 64, 96, 128, 256]))
        S = int(rng.choice([2, 16, 50, 64, 100, 256, 300]))
        T = int(rng.choice([3, 6, 13, 32, 50, 90]))
        B = int(rng.choice([1100, 3000, 8193, 8257, 9000, 12345, 16384, 20000]))
        pad = float(rng.choice([0.0, 0.5, 0.98]))
        params = model_params(mode, V, E, Hs, Ht, S, T)
        m, p = make_pair(params, seed=int(rng.randint(1 << 30)))
        h = m.handle
        h.set_option("lstm_small_rows", 0)
        h.set_option("lstm_cluster_rows", 0)
        ids = random_ids(rng, B, T, V, pad_frac=pad)
        if pad > 0:
            ids[rng.randint(B)] = 0                                     # an all-PAD row
            r = rng.randint(B)
            ids[r, rng.randint(T)] = 0                                   # a PAD inside a row
            dense = rng.choice(B, B // 10, replace=False)
            ids[dense] = random_ids(rng, len(dense), T, V)               # mixed: a tenth of the rows without padding
        tag = "%s V=%d E=%d Hs=%d Ht=%d S=%d T=%d B=%d pad=%.2f" % (mode, V, E, Hs, Ht, S, T, B, pad)
        d = torch.from_numpy(ids).to(dev)
        out = torch.empty((B, S), dtype=torch.float32, device=dev)
        ok = True
        for side, name in ((0, "src"), (1, "tgt")):
            ref = None
            for gs in (1, 0):
                for sort in (1, 2, 0, 1):
                    h.set_option("lstm_gate_split", gs)
                    h.set_option("pad_sort_dev", sort)
                    out.fill_(7.0)
                    h.encode_dev(side, d.data_ptr(), B, T, True, out.data_ptr())
                    h.synchronize()
                    got = out.cpu().numpy()
                    if ref is None:
                        ref = got.copy()
                    elif not np.array_equal(got, ref):
                        ok = False
                        print("BITS DIFFER: %s side %s gate_split %d pad_sort_dev %d: %d rows" % (tag, name, gs, sort, int((got != ref).any(axis=1).sum())))
            pick = rng.choice(B, 48, replace=False)
            err = float(np.abs(ref[pick] - O.encode(p, params, name, ids[pick])).max())
            worst = max(worst, err)
            if err > 1e-4:
                ok = False
                print("ORACLE: %s side %s max |diff| %.2e" % (tag, name, err))
        h.close()
        failures += 0 if ok else 1
        if verbose:
            print("%3d %s  %s" % (case, "ok  " if ok else "FAIL", tag), flush=True)
    if verbose:
        print("%d cases, %d failures, worst |encoding - oracle| %.2e, %.0f s" % (n_cases, failures, worst, time.time() - t_start))
    return failures, worst


if __name__ == "__main__":
    f, _ = main(int(sys.argv[1]) if len(sys.argv) > 1 else 12, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    sys.exit(1 if f else 0)

#!/usr/bin/env python
"""SURVEY 8d measurement sweeps on one MI355X (inputs resident in HBM):
  C2  dual-encoder H=S=256 E=50 T=32 V=32000: encoded seqs/s for B in {1, 64, 1024, 16384, 131072}, both encoders;
  C4  synthetic ranking, S=256: 100,000 queries x 10,000,000 targets, k=10, as 8 LOGICAL shards of 1.25 M rows
      scored one after the other on this GPU (id_base = shard offset) + the k-way merge the RCCL all-gather feeds;
      top-1 checked against planted targets (normalize(q + 0.1*noise) at row j of shard j%8).
usage: tools/bench_configs.py [c2] [c4]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import sse_amd  # noqa: E402

which = set(sys.argv[1:]) or {"c2", "c4", "shapes"}
dev = torch.device("cuda:0")
V, E, H, S, T = 32000, 50, 256, 256, 32
params = dict(forward_only=True, network_mode="dual-encoder", predict_nbest=10, max_seq_length=T, vocab_size=V,
              embedding_size=E, encoding_size=S, src_cell_size=H, tgt_cell_size=H, learning_rate=0.9,
              learning_rate_decay_factor=0.99, targetSpaceSize=571)
m = sse_amd.SSEModel(params)
m.init_variables(seed=0)
h = m.handle
FLOP = T * 8 * H * (E + H) + 2 * H * S

if "c2" in which:
    for B in (1, 64, 1024, 16384, 131072):
        ids = torch.randint(2, V, (B, T), device=dev, dtype=torch.int32)
        ids[:, -1] = 1
        out = torch.empty((B, S), device=dev)
        for side, name in ((0, "src"), (1, "tgt")):
            for _ in range(3):
                h.encode_dev(side, ids.data_ptr(), B, T, True, out.data_ptr())
            n = 200 if B <= 1024 else 20 if B <= 16384 else 5
            h.timer_record(0)
            for _ in range(n):
                h.encode_dev(side, ids.data_ptr(), B, T, True, out.data_ptr())
            h.timer_record(1)
            ms = h.timer_elapsed_ms(0, 1) / n
            print("C2 %s B=%-6d %.3f ms/call  %.0f seq/s  %.1f TFLOP/s (%.1f%% of fp32 MFMA peak)"
                  % (name, B, ms, B / ms * 1e3, B * FLOP / ms / 1e9, B * FLOP / ms / 1e9 / 157.3 * 100))

if "shapes" in which:
    # other model shapes at B = 16384: the reference's defaults (sse_train.py:60-74: E=50, H=96, S=64, T=80),
    # configs[0] (shared-encoder H=128), the crosslingual recipe shape, and a wide cell (H=512: inference only)
    for (mode, E2, H2, S2, T2) in (("dual-encoder", 50, 96, 64, 80), ("shared-encoder", 50, 128, 64, 80),
                                   ("dual-encoder", 40, 50, 50, 50), ("dual-encoder", 50, 256, 256, 50),
                                   ("dual-encoder", 50, 512, 512, 32)):
        p2 = dict(params, network_mode=mode, embedding_size=E2, src_cell_size=H2, tgt_cell_size=H2, encoding_size=S2,
                  max_seq_length=T2)
        m2 = sse_amd.SSEModel(p2)
        m2.init_variables(seed=0)
        B = 16384
        ids = torch.randint(2, V, (B, T2), device=dev, dtype=torch.int32)
        out = torch.empty((B, S2), device=dev)
        for _ in range(2):
            m2.handle.encode_dev(0, ids.data_ptr(), B, T2, True, out.data_ptr())
        m2.handle.timer_record(0)
        for _ in range(10):
            m2.handle.encode_dev(0, ids.data_ptr(), B, T2, True, out.data_ptr())
        m2.handle.timer_record(1)
        ms = m2.handle.timer_elapsed_ms(0, 1) / 10
        fl = T2 * 8 * H2 * (E2 + H2) + 2 * H2 * S2
        print("shape %s E=%d H=%d S=%d T=%d B=%d: %.3f ms  %.0f seq/s  %.1f TFLOP/s algorithmic (%.1f%% of fp32 MFMA peak)"
              % (mode, E2, H2, S2, T2, B, ms, B / ms * 1e3, B * fl / ms / 1e9, B * fl / ms / 1e9 / 157.3 * 100))
        m2.handle.close()

if "c4" in which:
    Q, NS, P, k = 100000, 1250000, 8, 10
    gq = torch.Generator(device=dev).manual_seed(2)
    q = torch.nn.functional.normalize(torch.randn((Q, S), generator=gq, device=dev), dim=1)
    noise = torch.randn((Q, S), generator=gq, device=dev)
    all_s = torch.empty((P, Q, k), dtype=torch.float64, device=dev)
    all_i = torch.empty((P, Q, k), dtype=torch.int64, device=dev)
    jj = torch.arange(Q, device=dev)
    t_build = t_score = 0.0
    for p in range(P):
        g = torch.Generator(device=dev).manual_seed(100 + p)
        shard = torch.nn.functional.normalize(torch.randn((NS, S), generator=g, device=dev), dim=1)
        mine = jj[jj % P == p]
        shard[mine] = torch.nn.functional.normalize(q[mine] + 0.1 * noise[mine], dim=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        h.index_set_dev(shard.data_ptr(), NS, S, id_base=p * NS)
        h.synchronize()
        t1 = time.perf_counter()
        h.score_topk_dev(q.data_ptr(), Q, k, all_s[p].data_ptr(), all_i[p].data_ptr())
        h.synchronize()
        t2 = time.perf_counter()
        t_build += t1 - t0
        t_score += t2 - t1
        del shard
    out_s = torch.empty((Q, k), dtype=torch.float64, device=dev)
    out_i = torch.empty((Q, k), dtype=torch.int64, device=dev)
    t0 = time.perf_counter()
    h.merge_topk_dev(all_s.data_ptr(), all_i.data_ptr(), P, Q, k, out_s.data_ptr(), out_i.data_ptr())
    h.synchronize()
    t_merge = time.perf_counter() - t0
    acc = float((out_i[:, 0] == (jj % P) * NS + jj).double().mean().item())
    srt = bool((out_s[:, :-1] >= out_s[:, 1:]).all().item())
    tot = t_score + t_merge
    print("C4 %d queries x %d targets (8 logical shards on ONE GPU): scoring %.3f s + merge %.4f s = %.3f s  "
          "%.3e scores/s  %.1f TFLOP/s algorithmic (library default: candidates on the bf16 matrix pipe, exact float64 "
          "re-scoring; fp32 MFMA peak is 157.3); index layout build %.3f s; planted top-1 acc %.4f; "
          "rows sorted %s" % (Q, NS * P, t_score, t_merge, tot, Q * NS * P / tot, 2.0 * S * Q * NS * P / tot / 1e12,
                              t_build, acc, srt))

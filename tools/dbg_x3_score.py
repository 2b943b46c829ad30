import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import sse_amd
V, E, H, S, T, N, B = 32000, 50, 256, 256, 32, 571, 16384
params = dict(forward_only=True, network_mode="dual-encoder", predict_nbest=10, max_seq_length=T, vocab_size=V,
              embedding_size=E, encoding_size=S, src_cell_size=H, tgt_cell_size=H, learning_rate=0.9,
              learning_rate_decay_factor=0.99, targetSpaceSize=N)
m = sse_amd.SSEModel(params); m.init_variables(seed=0); h = m.handle
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1234)
tgt_ids = torch.randint(2, V, (N, T), generator=g, device=dev, dtype=torch.int32); tgt_ids[:, -1] = 1
tgt_enc = torch.empty((N, S), device=dev)
h.encode_dev(1, tgt_ids.data_ptr(), N, T, True, tgt_enc.data_ptr())
h.index_set_dev(tgt_enc.data_ptr(), N, S)
src_ids = torch.randint(2, V, (B, T), generator=g, device=dev, dtype=torch.int32); src_ids[:, -1] = 1
src_enc = torch.empty((B, S), device=dev)
top_s = torch.empty((B, 10), dtype=torch.float64, device=dev); top_i = torch.empty((B, 10), dtype=torch.int64, device=dev)
for x3 in (0, 1, 0, 1):
    h.set_option("lstm_x3", x3)
    for _ in range(2):
        h.encode_dev(0, src_ids.data_ptr(), B, T, True, src_enc.data_ptr())
        h.score_topk_dev(src_enc.data_ptr(), B, 10, top_s.data_ptr(), top_i.data_ptr())
    torch.cuda.synchronize()
    c0 = {k: h.get_counter(k) for k in ("score_bf16_second_chance_queries", "score_collect_queries", "score_bruteforce_queries")}
    t0 = time.perf_counter()
    for _ in range(10):
        h.encode_dev(0, src_ids.data_ptr(), B, T, True, src_enc.data_ptr())
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(10):
        h.score_topk_dev(src_enc.data_ptr(), B, 10, top_s.data_ptr(), top_i.data_ptr())
    torch.cuda.synchronize(); t2 = time.perf_counter()
    c1 = {k: h.get_counter(k) for k in c0}
    print("x3=%d encode %.3f ms  score %.3f ms  counters +%s" % (x3, (t1 - t0) * 100, (t2 - t1) * 100, {k: c1[k] - c0[k] for k in c0}))

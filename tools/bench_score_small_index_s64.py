import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, sse_amd
for S in (64, 50):
    Q=16384; N=571
    params = dict(forward_only=True, network_mode="dual-encoder", predict_nbest=10, max_seq_length=4, vocab_size=50, embedding_size=8, encoding_size=S, src_cell_size=16, tgt_cell_size=16, learning_rate=0.9, learning_rate_decay_factor=0.99, targetSpaceSize=5)
    h = sse_amd.SSEModel(params).handle
    dev=torch.device("cuda:0")
    t = torch.nn.functional.normalize(torch.randn((N,S),device=dev),dim=1)
    q = torch.nn.functional.normalize(torch.randn((Q,S),device=dev),dim=1)
    h.index_set_dev(t.data_ptr(), N, S)
    s = torch.empty((Q,10),dtype=torch.float64,device=dev); i=torch.empty((Q,10),dtype=torch.int64,device=dev)
    for opt in (0, 1):
        h.set_option("score_small_index", opt)
        for _ in range(5): h.score_topk_dev(q.data_ptr(), Q, 10, s.data_ptr(), i.data_ptr())
        torch.cuda.synchronize()
        t0=time.perf_counter()
        for _ in range(20): h.score_topk_dev(q.data_ptr(), Q, 10, s.data_ptr(), i.data_ptr())
        torch.cuda.synchronize()
        print("16384 x 571 x %d, score_small_index=%d: %.3f ms" % (S, opt, (time.perf_counter()-t0)/20*1e3))

import os, sys, time
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/sse_amd.py") else os.getcwd())
import torch, sse_amd
S=256; Q=16384; N=571
params = dict(forward_only=True, network_mode="dual-encoder", predict_nbest=10, max_seq_length=4, vocab_size=50, embedding_size=8, encoding_size=S, src_cell_size=16, tgt_cell_size=16, learning_rate=0.9, learning_rate_decay_factor=0.99, targetSpaceSize=5)
h = sse_amd.SSEModel(params).handle
dev=torch.device("cuda:0")
t = torch.nn.functional.normalize(torch.randn((N,S),device=dev),dim=1)
q = torch.nn.functional.normalize(torch.randn((Q,S),device=dev),dim=1)
h.index_set_dev(t.data_ptr(), N, S)
s = torch.empty((Q,10),dtype=torch.float64,device=dev); i=torch.empty((Q,10),dtype=torch.int64,device=dev)
for _ in range(12): h.score_topk_dev(q.data_ptr(), Q, 10, s.data_ptr(), i.data_ptr())
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(20): h.score_topk_dev(q.data_ptr(), Q, 10, s.data_ptr(), i.data_ptr())
torch.cuda.synchronize()
print("16384 x 571 x 256 score call: %.3f ms" % ((time.perf_counter()-t0)/20*1e3))

#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r05g; mkdir -p $o
cat > /tmp/dwt.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch, sse_amd
V,E,S,T=32000,50,512,64
params=dict(forward_only=False,network_mode="source_only_cnn",predict_nbest=10,max_seq_length=T,vocab_size=V,embedding_size=E,encoding_size=S,src_cell_size=96,tgt_cell_size=96,learning_rate=0.9,learning_rate_decay_factor=0.99,targetSpaceSize=571)
m=sse_amd.SSEModel(params); m.init_variables(seed=0); m.handle.set_option("cnn_bf16",1)
rng=np.random.RandomState(0); Bt=8192
src=np.repeat(rng.randint(2,V,size=(Bt//2,T)).astype(np.int32),2,axis=0); rows=rng.randint(0,571,size=Bt).astype(np.int32); z=np.tile(np.array([1.0,0.0],np.float32),Bt//2)
for _ in range(3): m.train_step(src,rows,z)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(10): m.train_step(src,rows,z)
torch.cuda.synchronize(); print("SSE_DW_DBG=%s step %.3f ms" % (os.environ.get("SSE_DW_DBG","0"), (time.perf_counter()-t0)/10*1e3))
PY
for d in 0 1 2 4 3 6 7; do SSE_DW_DBG=$d python /tmp/dwt.py 2>&1 | grep step; done | tee $o/dw_dbg.txt

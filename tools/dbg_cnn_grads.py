import sys, numpy as np, torch
sys.path.insert(0,'/root/repo')
from oracle import sse_oracle as O
from tests.util import make_pair, model_params, random_ids
import sse_amd
rng=np.random.RandomState(int(sys.argv[1]) if len(sys.argv)>1 else 0)
V,E,S,T,B,N=500,64,16,80,64,33
params=model_params("source_only_cnn",V,E,96,256,S,T,N=N,lr=0.5)
m,p=make_pair(params,seed=5)
src=random_ids(rng,B,T,V,0.0); rows=rng.randint(0,N,size=B).astype(np.int32); z=np.tile(np.array([1.0,0.0],np.float32),B//2)
loss,acc,grads=O.gradients(p,params,src,rows,z)
n=m.handle.train_grad_count(); a=torch.zeros(n,device='cuda:0'); m.handle.train_bind_arena(a)
m.handle.train_grads(src,rows,z)
torch.cuda.synchronize(); arena=a.cpu().numpy()
names=[v[0] for v in m.handle.variables() if not v[0].endswith('/Adagrad')]
print(names)
off=0
for i,name in enumerate(names):
    w=p[name]; g=arena[off:off+w.size].reshape(w.shape); off+=w.size
    og=grads[name]
    if isinstance(og,tuple): og=O.dense_embedding_grad(og,w.shape[0])
    d=np.abs(g-og); print(name, "max|g|=%.3e max diff=%.3e"%(np.abs(og).max(), d.max()), "argmax", np.unravel_index(d.argmax(), d.shape))
print("tail",arena[-4:], "oracle gnorm^2 slices", sum(float(np.sum(np.square(g[1],dtype=np.float64))) for g in grads.values() if isinstance(g,tuple)), float(loss))

#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=$GRAFT_REPO_ROOT/gpurun_out/r05h; mkdir -p $o
cat > /tmp/dwt.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch, sse_amd
V,E,S,T=32000,50,512,64
params=dict(forward_only=False,network_mode="source_only_cnn",predict_nbest=10,max_seq_length=T,vocab_size=V,embedding_size=E,encoding_size=S,src_cell_size=96,tgt_cell_size=96,learning_rate=0.9,learning_rate_decay_factor=0.99,targetSpaceSize=571)
m=sse_amd.SSEModel(params); m.init_variables(seed=0); m.handle.set_option("cnn_bf16",1)
rng=np.random.RandomState(0); Bt=8192
src=np.repeat(rng.randint(2,V,size=(Bt//2,T)).astype(np.int32),2,axis=0); rows=rng.randint(0,571,size=Bt).astype(np.int32); z=np.tile(np.array([1.0,0.0],np.float32),Bt//2)
for _ in range(4): m.train_step(src,rows,z)
torch.cuda.synchronize()
PY
cd /tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM" "SQ_IFETCH SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $o/p$i -o p -- python /tmp/dwt.py > $o/log$i.txt 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(list)
for f in glob.glob('gpurun_out/r05h/p*/p_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"].replace("(anonymous namespace)::","")
        if any(k in n for k in ("cnn_dw_kernel","cnn_dx_mfma","conv_pool_bf16")):
            agg[(n.split("(")[0][:40], r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()): print("%-42s grid=%-8s %-28s n=%d avg=%.5g" % (k[0],k[1],k[2],len(v),sum(v)/len(v)))
PY
find $o -name "*.csv" -size +5M -delete; rm -rf $o/p*/

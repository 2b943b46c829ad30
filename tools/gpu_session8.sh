#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
o=$GRAFT_REPO_ROOT/gpurun_out/s8; mkdir -p $o; rm -f $o/train.txt
echo "== current, unpaired" >> $o/train.txt
SSE_TRAIN_UNPAIRED=1 timeout 300 python tools/bench_train.py 128 1024 8192 >> $o/train.txt 2>&1
echo "== current, paired" >> $o/train.txt
timeout 300 python tools/bench_train.py 128 1024 8192 >> $o/train.txt 2>&1
cat $o/train.txt | grep -v amdgpu.ids
cd /tmp
SSE_TRAIN_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $o/t1024 -o p -- python $GRAFT_REPO_ROOT/tools/bench_train.py 1024 > $o/t1024.txt 2>&1
f=$(find $o/t1024 -name "*kernel_stats.csv" | head -1)
head -24 "$f" | cut -c1-140
find $o -name "*.csv" -size +20M -delete

#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
o=$GRAFT_REPO_ROOT/gpurun_out/s7; mkdir -p $o
cd /tmp
SSE_TRAIN_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $o/train_stats -o p -- python $GRAFT_REPO_ROOT/tools/bench_train.py 8192 > $o/train_serial.txt 2>&1
tail -2 $o/train_serial.txt
f=$(find $o/train_stats -name "*kernel_stats.csv" | head -1)
head -14 "$f" | cut -c1-150
find $o -name "*.csv" -size +20M -delete

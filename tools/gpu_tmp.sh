#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
o=$GRAFT_REPO_ROOT/gpurun_out/s21; mkdir -p $o
timeout 600 python tools/fuzz_parity.py 300 555 2>&1 | grep -v "^ok" | grep -v amdgpu | tail -4
cd /tmp
SSE_TRAIN_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $o/train_stats -o p -- python $GRAFT_REPO_ROOT/tools/bench_train.py 8192 > $o/train_serial.txt 2>&1
head -12 $(find $o/train_stats -name "*kernel_stats.csv" | head -1) | cut -c1-140
rocprofv3 --kernel-trace --output-format csv -d $o/tdef -o p -- python $GRAFT_REPO_ROOT/tools/bench_train_default.py 128 > $o/tdef.txt 2>&1
python $GRAFT_REPO_ROOT/tools/analyze_step_trace.py $(find $o/tdef -name "*kernel_trace.csv" | head -1) | grep -v "    [0-9.]* *[0-9.]* idle *0.0  q0   \(adagrad\|sumsq\|pack\)" | tail -40
find $o -name "*.csv" -size +20M -delete

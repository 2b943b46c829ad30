#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
o=$GRAFT_REPO_ROOT/gpurun_out/s9; mkdir -p $o
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $o/tdef -o p -- python $GRAFT_REPO_ROOT/tools/bench_train_default.py 128 > $o/tdef.txt 2>&1
tail -1 $o/tdef.txt
python $GRAFT_REPO_ROOT/tools/analyze_step_trace.py $(find $o/tdef -name "*kernel_trace.csv" | head -1) > $o/tdef_timeline.txt
cat $o/tdef_timeline.txt
find $o -name "*.csv" -size +20M -delete

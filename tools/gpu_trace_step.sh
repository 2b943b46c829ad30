#!/bin/bash
# kernel timeline of one concurrent train step (rocprofv3 kernel trace of tools/bench_train.py <rows>, or of
# tools/bench_train_default.py <rows> when the first argument is "default")
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
script=tools/bench_train.py
if [ "$1" = default ]; then script=tools/bench_train_default.py; shift; fi
rows=${1:-8192}; o=gpurun_out/trace_$rows; mkdir -p $o; root=$(pwd)
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $root/$o -o p -- python $root/$script $rows 2>&1 | grep B_rows )
python tools/analyze_step_trace.py $o/p_kernel_trace.csv 2 | tail -${2:-45}

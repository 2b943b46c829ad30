#!/bin/bash
# kernel timeline of one concurrent train step (rocprofv3 kernel trace of tools/bench_train.py <rows>)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
rows=${1:-8192}; o=gpurun_out/trace_$rows; mkdir -p $o; root=$(pwd)
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $root/$o -o p -- python $root/tools/bench_train.py $rows 2>&1 | grep B_rows )
python tools/analyze_step_trace.py $o/p_kernel_trace.csv 2 | grep -v "idle    0.0  q.* \(pack\|pad_rows\|fill\)" | tail -45

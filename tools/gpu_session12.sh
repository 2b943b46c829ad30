#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
o=$GRAFT_REPO_ROOT/gpurun_out/s12; mkdir -p $o
timeout 2400 python -m pytest tests -m gpu -q --durations=8 > $o/tests.log 2>&1; echo "rc=$?" >> $o/tests.log; tail -16 $o/tests.log
timeout 300 python tools/bench_demo_query.py 10000000 > $o/demo.txt 2>&1; grep -v amdgpu.ids $o/demo.txt | tail -28

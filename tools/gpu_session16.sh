#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
o=$GRAFT_REPO_ROOT/gpurun_out/s16; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_encode.py -m gpu -q -x -s -k "split_bf16" > $o/tests.log 2>&1; echo "rc=$?" >> $o/tests.log; grep -v amdgpu.ids $o/tests.log | grep -v "^lstm_x3" | tail -8
timeout 300 python tools/bench_x3.py 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/bench_x3.py 16384 96 64 80 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/bench_x3.py 16384 128 64 80 2>&1 | grep -v amdgpu.ids
timeout 600 python bench.py --no-cpu-baseline > $o/bench.json 2> $o/bench.err; tail -1 $o/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], json.dumps(d.get('encode_leg_split_bf16')))"

#!/bin/bash
# round 6 call S: LDS flags as workgroup-scope atomics (ds_read / ds_write) instead of volatile generic accesses (flat_load + vmcnt(0))
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
timeout 300 python tools/bench_shapes.py 2>&1 | grep "E=" 
timeout 300 python tools/bench_train_default.py 2>&1 | grep -v amdgpu
timeout 300 python tools/bench_x3.py 2>&1 | grep -v amdgpu | tail -6
timeout 1200 python -m pytest tests/test_gpu_encode.py tests/test_gpu_train.py tests/test_gpu_fuzz.py tests/test_gpu_generic.py -x -q 2>&1 | tail -3

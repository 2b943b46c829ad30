#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
timeout 600 python -m pytest tests/test_gpu_score.py tests/test_gpu_rccl.py -x -q 2>&1 | tail -12 | cut -c1-250
timeout 300 python tools/bench_c3.py 2>&1 | grep -v amdgpu | cut -c1-250

#!/bin/bash
# round 6 call L: score + rccl + generic tests, then the whole suite with durations (which test makes the late ones slow?)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06l; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_score.py tests/test_gpu_rccl.py -x -q 2>&1 | tail -5
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -30 > $o/suite.txt; grep -v "^$" $o/suite.txt | cut -c1-250

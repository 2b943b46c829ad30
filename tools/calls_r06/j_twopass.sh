#!/bin/bash
# round 6 call J: two-pass mid-size scoring: tests, C3 timing A/B (tools/bench_c3.py), the whole suite with durations
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06j; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_score.py tests/test_gpu_generic.py -x -q 2>&1 | tail -8
timeout 300 python tools/bench_c3.py > $o/c3.txt 2>&1; tail -8 $o/c3.txt
timeout 1500 python -m pytest tests -m gpu -x -q --durations=25 2>&1 | tail -40 > $o/suite.txt; grep -v "^$" $o/suite.txt | cut -c1-200

#!/bin/bash
# round 6 call A: the default bench line (headline size check) + the bench-contract GPU tests
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06a; mkdir -p $o
timeout 400 python bench.py > $o/bench.out 2> $o/bench.err
tail -n 1 $o/bench.out > $o/bench_headline.json
wc -c $o/bench_headline.json
cp profiles/bench_legs_latest.json $o/bench_legs.json
timeout 600 python -m pytest tests/test_gpu_bench_contract.py -x -q 2>&1 | tail -5

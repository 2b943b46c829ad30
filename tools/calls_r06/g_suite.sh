#!/bin/bash
# round 6 call G: whole GPU suite + default bench line on the current sources
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06g; mkdir -p $o
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12
timeout 400 python bench.py > $o/bench.out 2> $o/bench.err
tail -n 1 $o/bench.out > $o/bench_headline.json; wc -c $o/bench_headline.json; cat $o/bench_headline.json
cp profiles/bench_legs_latest.json $o/bench_legs.json

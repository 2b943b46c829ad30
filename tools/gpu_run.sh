#!/bin/bash
# One gpurun call: the GPU test suite and / or a bench line, outputs under gpurun_out/<tag>/.
# usage (on the GPU box, repo root): bash tools/gpu_run.sh <tag> [tests] [bench] [-- extra pytest args]
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
tag=$1; shift
o=gpurun_out/$tag; mkdir -p $o
for what in "$@"; do
  case $what in
    tests) timeout 1500 python -m pytest tests -m gpu -q -x > $o/tests.log 2>&1; echo "rc=$?" >> $o/tests.log; tail -15 $o/tests.log ;;
    bench) timeout 900 python bench.py > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"; tail -c 600 $o/bench.err ;;
  esac
done

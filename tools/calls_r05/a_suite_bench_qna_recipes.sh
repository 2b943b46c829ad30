#!/bin/bash
# round 5, call A: the whole GPU suite (no -x: every failure in one call), the default bench line, the qna recipe runs
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r05a; mkdir -p $o
timeout 1500 python -m pytest tests -m gpu -q --timeout 700 -p no:cacheprovider --durations=15 > $o/tests.log 2>&1; echo "tests rc=$?" | tee -a $o/tests.log
grep -n "passed\|failed\|error" $o/tests.log | tail -5
( time timeout 600 python bench.py > $o/bench.json 2> $o/bench.err ) 2> $o/bench.time; echo "bench rc=$?"; tail -3 $o/bench.time; tail -c 400 $o/bench.err
timeout 400 python tools/train_recipe_from_ids.py qna --epochs 40 --lr 0.9 > $o/recipe_qna_lr0.9.txt 2>&1; echo "qna rc=$?"; tail -3 $o/recipe_qna_lr0.9.txt
timeout 400 python tools/train_recipe_from_ids.py qna --epochs 40 --lr 0.005 > $o/recipe_qna_lr0.005.txt 2>&1; echo "qna2 rc=$?"; tail -3 $o/recipe_qna_lr0.005.txt

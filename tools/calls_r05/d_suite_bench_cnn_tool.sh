#!/bin/bash
# round 5: the whole GPU suite + the default bench line + the CNN tool on the current sources
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r05d; mkdir -p $o
timeout 1500 python -m pytest tests -m gpu -q --timeout 700 -p no:cacheprovider --durations=8 > $o/tests.log 2>&1; echo "tests rc=$?" | tee -a $o/tests.log
grep -n "passed\|failed\|^FAILED" $o/tests.log | tail -8
( time timeout 600 python bench.py > $o/bench.json 2> $o/bench.err ) 2> $o/bench.time; echo "bench rc=$?"; tail -3 $o/bench.time
python tools/bench_cnn.py 2>&1 | tail -6 | tee $o/cnn.txt

#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r05fuzz; mkdir -p $o
timeout 200 python -m pytest tests/test_gpu_generic.py -m gpu -q --timeout 150 -p no:cacheprovider > $o/tests.log 2>&1; echo "tests rc=$?"; grep -n "passed\|failed\|^FAILED" $o/tests.log | tail -3
timeout 300 python tools/fuzz_parity.py 160 505 > $o/fuzz.txt 2>&1; echo "rc=$?"; grep "^FAIL\|fuzz summary" $o/fuzz.txt | head -12
timeout 300 python tools/fuzz_parity.py 200 77 > $o/fuzz2.txt 2>&1; echo "rc=$?"; grep "^FAIL\|fuzz summary" $o/fuzz2.txt | head -12

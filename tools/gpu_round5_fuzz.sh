#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r05fuzz; mkdir -p $o
timeout 420 python tools/fuzz_parity.py 160 505 > $o/fuzz.txt 2>&1; echo "rc=$?"; grep -c "^ok" $o/fuzz.txt; grep "^FAIL\|fuzz summary" $o/fuzz.txt | head -20

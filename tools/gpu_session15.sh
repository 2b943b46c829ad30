#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
o=$GRAFT_REPO_ROOT/gpurun_out/s15; mkdir -p $o
timeout 900 python tools/fuzz_parity.py 120 7 > $o/fuzz.txt 2>&1; grep -v "^ok" $o/fuzz.txt | grep -v amdgpu.ids | tail -15

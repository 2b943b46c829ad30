#!/bin/bash
# round 6 call X: randomised sweeps on the final sources
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06x_fuzz; mkdir -p $o
for seed in 21 22 23; do timeout 900 python tools/fuzz_parity.py 60 $seed 2>&1 | grep -v amdgpu | tail -1 | tee -a $o/fuzz.txt | cut -c1-250; done
for seed in 7 8; do timeout 900 python tools/fuzz_large_batches.py 30 $seed 2>&1 | grep -v amdgpu | grep -v " ok  " | tee -a $o/fuzz_large.txt | cut -c1-200; done

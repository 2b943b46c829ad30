#!/bin/bash
# A/B of the candidate sweep (tools/bench_score.py, 8192 x 1.25 M x 256) over several builds of the library in ONE gpurun call
# (boxes differ by a few per cent).  usage (GPU box, repo root): bash tools/ab_score.sh <tag> lib1.so lib2.so ...
export TMPDIR=/tmp
tag=$1; shift
o=gpurun_out/$tag; mkdir -p $o
for rep in 1 2; do
  for lib in "$@"; do
    SSE_BENCH_PASSES=20 SSE_HIP_LIB=$(pwd)/sequence-semantic-embedding_amd/$lib timeout 200 python tools/bench_score.py > $o/run.txt 2>&1
    echo "== $lib (run $rep): $(grep '^bf16 ' $o/run.txt | cut -c1-60) | identical $(grep -c 'identical to fp32 candidates: True' $o/run.txt)" | tee -a $o/ab.txt
    grep "score clock <4,1>" $o/run.txt | grep "wave 0\|wave 4" | tail -2 | sed 's/\[score clock <4,1> KG=16\] //' | cut -c1-220 | tee -a $o/ab.txt
  done
done

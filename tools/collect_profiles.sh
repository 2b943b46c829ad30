#!/bin/bash
# Run on the GPU box from the repo root: bench line + rocprofv3 kernel stats + PMC passes (each counter group in its
# own run, kernel-trace only, as MI355X_MICROARCH.md prescribes), all under gpurun_out/<tag>/.
# usage: tools/collect_profiles.sh <tag>;  then: python tools/summarize_profiles.py <tag> gpurun_out/<tag>/stats gpurun_out/<tag>/pmc_*
set -u
exec </dev/null
tag=${1:-prof}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout 300 python bench.py > "$out/bench.json" 2> "$out/bench.err"
cd /tmp
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o p -- \
    python "$root/bench.py" > "$out/stats.log" 2>&1
groups=(FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE")
# PMC_MIN=1: only the three passes bench.py's roofline objects read (GPU-minutes); default: the wait / LDS / instruction mixes too
if [ -z "${PMC_MIN:-}" ]; then
  groups+=("SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
           "SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_IDX_ACTIVE"
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU")
fi
for c in "${groups[@]}"; do
    d="$out/pmc_$(echo $c | tr ' ' '_' | cut -c1-40)"
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$d" -o p -- \
        python "$root/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --score-iters 1 --no-shapes-leg --no-realdata-leg --no-sweep-leg > "$out/log_$(basename $d).txt" 2>&1
done
cd "$root"
find "$out" -name "*.csv" -size +20M -delete
ls -la "$out"

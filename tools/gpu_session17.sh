#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
o=$GRAFT_REPO_ROOT/gpurun_out/s17; mkdir -p $o
cd /tmp
for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; do
  d="$o/pmc_$(echo $c | tr ' ' '_' | cut -c1-40)"
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$d" -o p -- python $GRAFT_REPO_ROOT/tools/bench_x3.py > /dev/null 2>&1
  python - "$d/p_counter_collection.csv" <<'PY'
import csv,sys,collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'x3_kernel' in r['Kernel_Name'] or 'lstm_fwd_kernel<2, 2' in r['Kernel_Name']:
        agg[(r['Kernel_Name'][:34],r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print(k, 'n=%d avg=%.4g'%(len(v),sum(v)/len(v)))
PY
done

#!/bin/bash
# round 6 call W: the mid-batch cluster kernel: clock64 phases per step (measurement build -DSSE_LC_CLOCK) + MFMA-busy (rocprofv3 PMC)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06w; mkdir -p $o
SSE_HIP_LIB=$PWD/sequence-semantic-embedding_amd/libsse_lcclk.so N=3 timeout 300 python tools/dbg_cluster.py 600 1024 2>&1 | grep -v amdgpu | sort | uniq -c | sort -rn | head -12 | cut -c1-250 | tee $o/cluster_clock.txt
R=$PWD; ( cd /tmp && N=20 timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$o/pmc -o p -- python $R/tools/dbg_cluster.py 600 1024 > $R/$o/pmc.log 2>&1 )
python - <<PY
import csv, collections
busy, gui, dur = collections.defaultdict(list), collections.defaultdict(list), collections.defaultdict(list)
for r in csv.DictReader(open('$o/pmc/p_counter_collection.csv')):
    if 'lstm_cluster' not in r['Kernel_Name']: continue
    k = r['Grid_Size']
    if r['Counter_Name'] == 'SQ_VALU_MFMA_BUSY_CYCLES':
        busy[k].append(float(r['Counter_Value'])); dur[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    elif r['Counter_Name'] == 'GRBM_GUI_ACTIVE': gui[k].append(float(r['Counter_Value']))
for k in busy:
    b = sum(busy[k]) / len(busy[k]) / 1024.0; g = sum(gui[k]) / len(gui[k]) / 8.0
    print('lstm_cluster_kernel grid', k, 'n', len(busy[k]), 'MFMA-busy %.3f' % (b / g), 'avg dispatch %.3f ms' % (sum(dur[k]) / len(dur[k]) / 1e6))
PY

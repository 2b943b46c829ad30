#!/bin/bash
# Per-kernel register / scratch / LDS usage of one HIP source as the gfx950 compiler reports it
# (hipcc -Rpass-analysis=kernel-resource-usage; cross-compiles without a GPU).
#   tools/kernel_resources.sh sequence-semantic-embedding_amd/csrc/lstm_fwd.hip
set -e
src="$1"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c --cuda-device-only -Rpass-analysis=kernel-resource-usage \
  -o /dev/null "$src" 2>&1 | python3 -c '
import re, sys, subprocess
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"remark: (?:\s*)(Function Name|VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m: continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    elif cur is not None:
        cur[k.split(" ")[0]] = v
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.split("\n")
print("%-78s %5s %5s %5s %7s %4s %7s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "occ", "LDS"))
for r, n in zip(rows, names):
    n = re.sub(r"\(.*\)$", "", n).replace("void ", "")
    print("%-78s %5s %5s %5s %7s %4s %7s" % (n[:78], r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs"), r.get("ScratchSize"), r.get("Occupancy"), r.get("LDS")))
'

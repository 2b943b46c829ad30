#!/usr/bin/env python
"""Condense gpurun_out/<tag>/ (tools/collect_profiles_extra.sh) into tracked summaries under profiles/:
<tag>_<tool>.txt (the tool's own output), <tag>_<tool>_kernel_stats.csv (rocprofv3 --stats rows of library kernels),
<tag>_hbm_pmc.txt (FETCH_SIZE / WRITE_SIZE per dispatch for the HBM-bound kernels)."""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKIP = ("at::native", "rocclr", "Cijk", "hipcub", "rocprim", "elementwise", "distribution")


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    for txt in sorted(glob.glob(os.path.join(src, "*.txt"))):
        name = os.path.basename(txt)[:-4]
        lines = [l for l in open(txt, errors="replace") if "amdgpu.ids" not in l]
        open(os.path.join(dst, "%s_%s.txt" % (tag, name)), "w").writelines(lines)
        stats = os.path.join(src, name + "_stats", "p_kernel_stats.csv")
        if os.path.exists(stats):
            rows = list(csv.DictReader(open(stats)))
            with open(os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, name)), "w") as f:
                f.write("# rocprofv3 --kernel-trace --stats -- %s (library kernels; torch helper kernels dropped)\n" % name)
                w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
                w.writeheader()
                for r in rows:
                    if not any(k in r["Name"] for k in SKIP):
                        w.writerow(r)
    out = ["# rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace (separate passes); KB per dispatch, averaged by (kernel, grid).",
           "# gfx950: FETCH_SIZE counts half of the bytes of wide (16 B/lane) streaming reads (MI355X_MICROARCH.md) -> HBM read",
           "# bytes ~ 2 x FETCH_SIZE KB x 1024 for these kernels."]
    for d in sorted(glob.glob(os.path.join(src, "*_pmc_*"))):
        f = os.path.join(d, "p_counter_collection.csv")
        if not os.path.exists(f):
            continue
        agg, dur = collections.defaultdict(list), collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if any(k in r["Kernel_Name"] for k in SKIP):
                continue
            key = (r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:60], r["Grid_Size"], r["Counter_Name"])
            agg[key].append(float(r["Counter_Value"]))
            dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        # only the byte counters belong in this file (cycle counters of the MFMA-busy passes were listed here
        # under the "KB" header); FETCH_SIZE / WRITE_SIZE count kilobytes, the gfx950 correction is in the header above
        if not any(c in ("FETCH_SIZE", "WRITE_SIZE") for (_k, _g, c) in agg):
            continue
        out.append("## " + os.path.basename(d))
        for (k, g, c), v in sorted(agg.items()):
            if c not in ("FETCH_SIZE", "WRITE_SIZE") or sum(v) / len(v) < 1000:          # < 1 MB: not an HBM-bound launch
                continue
            avg_kb, avg_ms = sum(v) / len(v), sum(dur[(k, g, c)]) / len(v) / 1e6
            gbps = (2.0 if c == "FETCH_SIZE" else 1.0) * avg_kb * 1024 / (avg_ms * 1e-3) / 1e9
            out.append("%-60s grid=%-9s %-10s n=%-3d avg=%.5g KB  (avg dispatch %.3f ms; %s %.0f GB/s)"
                       % (k, g, c, len(v), avg_kb, avg_ms, "read ~2x:" if c == "FETCH_SIZE" else "write:", gbps))
    if len(out) > 3:
        open(os.path.join(dst, "%s_hbm_pmc.txt" % tag), "w").write("\n".join(out) + "\n")
    print("\n".join(out[-40:]))
    # MFMA-busy share of the training kernels: (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs) per dispatch
    for d in sorted(glob.glob(os.path.join(src, "*_pmc_SQ_VALU*"))):
        tool = os.path.basename(d).split("_pmc_")[0]
        f = os.path.join(d, "p_counter_collection.csv")
        if not os.path.exists(f):
            continue
        busy, gui, dur = collections.defaultdict(list), collections.defaultdict(list), collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if any(k in r["Kernel_Name"] for k in SKIP):
                continue
            key = (r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:60], r["Grid_Size"])
            if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
                busy[key].append(float(r["Counter_Value"]))
                dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                gui[key].append(float(r["Counter_Value"]))
        lines = ["# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- the '%s' command of tools/collect_profiles_extra.sh" % tool,
                 "# MFMA-busy = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs), averaged over the dispatches of a (kernel, grid)"]
        for key in sorted(busy, key=lambda k: -sum(dur[k])):
            if key not in gui or sum(dur[key]) / len(dur[key]) < 20000:
                continue
            b = sum(busy[key]) / len(busy[key]) / 1024.0
            g = sum(gui[key]) / len(gui[key]) / 8.0
            lines.append("%-60s grid=%-9s n=%-3d MFMA-busy %.3f  (avg dispatch %.3f ms)"
                         % (key[0], key[1], len(busy[key]), b / g if g else 0.0, sum(dur[key]) / len(dur[key]) / 1e6))
        open(os.path.join(dst, "%s_%s_pmc.txt" % (tag, tool)), "w").write("\n".join(lines) + "\n")
        print("\n".join(lines))


if __name__ == "__main__":
    main()

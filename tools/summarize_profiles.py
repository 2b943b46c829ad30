#!/usr/bin/env python
"""Condense rocprofv3 CSV outputs under gpurun_out/ into the small tracked summaries under profiles/.
usage: tools/summarize_profiles.py <tag> <stats_dir> <pmc_dir> [<pmc_dir> ...]"""
import collections
import csv
import os
import sys

OURS = ("score_small_index", "lstm_fwd", "lstm_bwd", "dk_x3", "dx_scatter", "emb_grad", "dx_hot", "compact_uncert", "gather_query", "multi_kernel", "proj_bwd_dm_mfma", "score_topk", "rescore", "exact_topk", "pack_", "merge_topk", "row_norm2", "pad_rows",
        "l2_normalize", "conv_pool", "proj_norm", "cnn_d", "cnn_bwd", "dk_gemm", "dx_kernel", "loss_", "adagrad", "sumsq", "clip_scale",
        "proj_bwd", "db_reduce", "dk_reduce")


def main():
    tag, stats_dir, pmc_dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    rows = list(csv.DictReader(open(os.path.join(stats_dir, "p_kernel_stats.csv"))))
    with open(os.path.join(root, "%s_kernel_stats.csv" % tag), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py"
                "  (library kernels only; torch setup kernels dropped)\n")
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in rows:
            if any(k in r["Name"] for k in OURS):
                w.writerow(r)
    # The default line's extra legs launch the dominant kernel at other sizes too (config_sweep: 1 .. 131072 rows, realdata_leg:
    # T = 50), so its all-calls average in the table above is not the headline launch's: per (kernel, grid) averages from the
    # kernel trace of the same run, largest total first.
    trace = os.path.join(stats_dir, "p_kernel_trace.csv")
    if os.path.exists(trace):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(trace)):
            n = r["Kernel_Name"].replace("(anonymous namespace)::", "")
            if any(k in n for k in ("lstm_fwd_kernel<2, 2, 1, false", "score_topk_kernel<4, true", "cnn_dx_mfma", "cnn_dw_kernel", "conv_pool_bf16", "proj_norm_x3")):
                per[(n.split("(")[0].replace("void ", ""), str(int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"])), r["Workgroup_Size_X"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        with open(os.path.join(root, "%s_kernel_by_grid.txt" % tag), "w") as f:
            f.write("# rocprofv3 --kernel-trace -- python bench.py: dispatch durations by (kernel, grid size in work-items, workgroup size)\n")
            f.write("# headline launch = lstm_fwd_kernel<2, 2, 1, false, ...> at grid 131072 (16384 sequences / 64 rows x 512 threads), T = 32\n")
            for (k, g, wg), v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
                f.write("%-58s grid=%-8s wg=%-4s calls=%-3d avg=%.4f ms  min=%.4f  max=%.4f\n"
                        % (k, g, wg, len(v), sum(v) / len(v) / 1e6, min(v) / 1e6, max(v) / 1e6))
    out = ["# rocprofv3 --pmc <counters> --kernel-trace (separate passes) -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --score-iters 1",
           "# per-dispatch averages by (kernel, grid); FETCH_SIZE/WRITE_SIZE in KB (MI355X_MICROARCH.md: FETCH_SIZE under-reports",
           "# wide 16 B/lane streaming reads 2x on gfx950); SQ_WAVE_CYCLES/SQ_WAIT_*/SQ_ACTIVE_* are quad-cycles summed over waves;",
           "# SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs"]
    for d in pmc_dirs:
        agg = collections.defaultdict(list)
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(os.path.join(d, "p_counter_collection.csv"))):
            if any(k in r["Kernel_Name"] for k in ("lstm_fwd", "lstm_bwd", "dk_x3", "dk_gemm", "dx_scatter", "conv_pool", "proj_norm", "cnn_d", "cnn_bwd", "lstm_cluster", "score_topk", "score_small_index", "rescore_kernel")):
                key = (r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:64], r["Grid_Size"])
                agg[key + (r["Counter_Name"],)].append(float(r["Counter_Value"]))
                dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for (k, g, c), v in sorted(agg.items()):
            out.append("%-56s grid=%-8s %-26s n=%-3d avg=%.5g  (avg dispatch %.3f ms)"
                       % (k, g, c, len(v), sum(v) / len(v), sum(dur[(k, g)]) / len(dur[(k, g)]) / 1e6))
    open(os.path.join(root, "%s_pmc.txt" % tag), "w").write("\n".join(out) + "\n")
    # Tracked summary bench.py quotes in its roofline objects -- only while the kernel sources are the ones profiled:
    #  * HBM bytes per launch of the dominant kernel = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE correction of
    #    MI355X_MICROARCH.md), full-size inference launches only;
    #  * MFMA-busy share per kernel = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs), largest-grid launches.
    import json
    sys.path.insert(0, os.path.dirname(root))
    import bench
    traffic, busy, gui, q1 = {}, collections.defaultdict(dict), collections.defaultdict(dict), []
    CNN = ("conv_pool", "proj_norm", "cnn_d", "cnn_bwd")
    cnn_io = collections.defaultdict(lambda: collections.defaultdict(dict))   # kernel -> counter -> grid -> [values]
    for d in pmc_dirs:
        for r in csv.DictReader(open(os.path.join(d, "p_counter_collection.csv"))):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()
            if ("lstm_fwd_kernel<2, 2, 1, false" in name and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE")
                    and int(r["Grid_Size"]) >= 65536):
                traffic.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            if "score_topk_kernel<1, true, false" in name and r["Counter_Name"] == "FETCH_SIZE" and float(r["Counter_Value"]) > 1e5:
                q1.append(float(r["Counter_Value"]))
            if any(k in name for k in CNN) and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                cnn_io[name][r["Counter_Name"]].setdefault(int(r["Grid_Size"]), []).append(float(r["Counter_Value"]))
            if r["Counter_Name"] in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
                tgt = busy if r["Counter_Name"].startswith("SQ") else gui
                tgt[name].setdefault(int(r["Grid_Size"]), []).append(float(r["Counter_Value"]))
    summary = {"csrc_sha": bench.csrc_sha(), "source": "profiles/%s_pmc.txt" % tag, "mfma_busy": {}}
    if "FETCH_SIZE" in traffic and "WRITE_SIZE" in traffic:
        f = sum(traffic["FETCH_SIZE"]) / len(traffic["FETCH_SIZE"])
        w_ = sum(traffic["WRITE_SIZE"]) / len(traffic["WRITE_SIZE"])
        summary["lstm_fwd_hbm_bytes_per_launch"] = (2 * f + w_) * 1024
        summary["lstm_fwd_fetch_size_kb"], summary["lstm_fwd_write_size_kb"] = f, w_
    if q1:
        summary["sweep_q1_hbm_bytes"] = 2 * sum(q1) / len(q1) * 1024
    for name in busy:
        g = max(busy[name])
        if g in gui.get(name, {}):
            b = sum(busy[name][g]) / len(busy[name][g]) / 1024.0
            a = sum(gui[name][g]) / len(gui[name][g]) / 8.0
            if a > 0 and (b / a > 0.005 or any(k in name for k in CNN)):   # (the CNN kernels are listed even at 0: "no MFMA" is a finding)
                summary["mfma_busy"][name] = round(b / a, 4)
    # configs[4] kernels: HBM bytes per launch of the largest grid = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (same correction)
    summary["cnn_hbm_bytes_per_launch"] = {}
    for name, io in cnn_io.items():
        if "FETCH_SIZE" in io and "WRITE_SIZE" in io:
            g = max(io["FETCH_SIZE"])
            if g in io["WRITE_SIZE"]:
                f = sum(io["FETCH_SIZE"][g]) / len(io["FETCH_SIZE"][g])
                w_ = sum(io["WRITE_SIZE"][g]) / len(io["WRITE_SIZE"][g])
                summary["cnn_hbm_bytes_per_launch"][name] = {"grid": g, "fetch_bytes": 2 * f * 1024, "write_bytes": w_ * 1024}
    json.dump(summary, open(os.path.join(root, "pmc_summary.json"), "w"), indent=1, sort_keys=True)
    print("\n".join(out))


if __name__ == "__main__":
    main()

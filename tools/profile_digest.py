#!/usr/bin/env python3
"""Digest of one tools/gpu_round2.sh / gpu_round3.sh session (gpurun_out/round2, gpurun_out/round3) into the tracked files under profiles/:

    python tools/profile_digest.py [gpurun_out/round2] [r02]

  <tag>_bench.json                 the bench line
  <tag>_bench_kernel_stats.csv     rocprofv3 --kernel-trace --stats of bench.py (steps 2, warmup 1)
  <tag>_loss_kernel_stats.csv      the same for tools/bench_loss.py
  <tag>_train_kernel_stats.csv     the same for tools/bench_config5.py --rows 65536 (large-batch training step)
  <tag>_train.jsonl                one line each: bench_train (32-row step), bench_config5 at 65,536 and 1,048,576 rows (fp32, then bf16), bench_train_kernels, bench_tft
  <tag>_pmc_forward.csv / .json    PMC passes over tools/bench_forward.py (--dtype bf16_fast, 3.65 M rows x 256 pairs)
  <tag>_pmc_train.csv              PMC passes over tools/bench_train_kernels.py (65,536 rows x 80 pairs: forward / backward, recomputing and saved)
  <tag>_pmc_reduce.csv / .json     PMC passes over tools/bench_reduce.py (262,144 rows x 8,192 pairs: a bench row block)
  <tag>_pmc_loss.json              PMC bytes per element and pass of mcmc_loss_fused at 3.65e7 and 4.56e6 elements (round 3 on)
HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB; gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md)."""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    return re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", ""))[-70:]


def stats_csv(src_dir, dst, top=14):
    files = glob.glob(os.path.join(src_dir, "**", "*kernel_stats.csv"), recursive=True)
    if not files:
        return False
    rows = list(csv.DictReader(open(max(files, key=os.path.getmtime))))  # gpurun merges sessions: take the newest
    with open(dst, "w") as f:
        f.write("kernel,calls,total_ms,avg_ms,min_ms,max_ms,percent\n")
        for r in rows[:top]:
            f.write(",".join([short(r["Name"]), r["Calls"], f"{int(r['TotalDurationNs']) / 1e6:.3f}", f"{float(r['AverageNs']) / 1e6:.4f}",
                              f"{int(r['MinNs']) / 1e6:.4f}", f"{int(r['MaxNs']) / 1e6:.4f}", r["Percentage"]]) + "\n")
    return True


def pmc(src_dir, filt):
    acc = defaultdict(lambda: [0.0, set()])
    newest = {}  # per pass directory the newest file (gpurun merges sessions into one tree)
    for f in glob.glob(os.path.join(src_dir, "pass*", "**", "*counter_collection.csv"), recursive=True):
        d = os.path.relpath(f, src_dir).split(os.sep)[0]
        if d not in newest or os.path.getmtime(f) > os.path.getmtime(newest[d]):
            newest[d] = f
    for f in sorted(newest.values()):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if filt not in k:
                continue
            a = acc[(k, r["Counter_Name"])]
            a[0] += float(r["Counter_Value"])
            a[1].add((f, r["Dispatch_Id"]))
    return {kc: v / len(d) for kc, (v, d) in acc.items()}


def write_pmc(avg, dst):
    with open(dst, "w") as f:
        f.write("kernel,counter,avg_per_dispatch\n")
        for (k, c), v in sorted(avg.items()):
            f.write(f"{k},{c},{v:.6g}\n")


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "round2")
    tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
    prof = os.path.join(ROOT, "profiles")
    if os.path.exists(os.path.join(src, "bench.json")):
        shutil.copy(os.path.join(src, "bench.json"), os.path.join(prof, f"{tag}_bench.json"))
    stats_csv(os.path.join(src, "prof"), os.path.join(prof, f"{tag}_bench_kernel_stats.csv"))
    stats_csv(os.path.join(src, "prof_loss"), os.path.join(prof, f"{tag}_loss_kernel_stats.csv"))
    stats_csv(os.path.join(src, "prof_train"), os.path.join(prof, f"{tag}_train_kernel_stats.csv"))
    if os.path.isdir(os.path.join(src, "prof_train_bf16")):
        stats_csv(os.path.join(src, "prof_train_bf16"), os.path.join(prof, f"{tag}_train_bf16_kernel_stats.csv"))
    if os.path.exists(os.path.join(src, "train.jsonl")):
        shutil.copy(os.path.join(src, "train.jsonl"), os.path.join(prof, f"{tag}_train.jsonl"))
    fwd = pmc(os.path.join(src, "pmc_fwd"), "forward_bf16")
    if fwd:
        write_pmc(fwd, os.path.join(prof, f"{tag}_pmc_forward.csv"))
        k = next(k for (k, c) in fwd if c == "FETCH_SIZE")
        units = 3_650_000 * 256
        hbm = (2 * fwd[(k, "FETCH_SIZE")] + fwd[(k, "WRITE_SIZE")]) * 1024.0
        busy, waves = fwd.get((k, "SQ_VALU_MFMA_BUSY_CYCLES")), fwd.get((k, "SQ_BUSY_CYCLES"))
        json.dump({"kernel": k, "launch": "3,650,000 rows x 256 pairs (tools/gpu_round2.sh: FETCH_SIZE / WRITE_SIZE in separate rocprofv3 passes)",
                   "FETCH_SIZE_KB": fwd[(k, "FETCH_SIZE")], "WRITE_SIZE_KB": fwd[(k, "WRITE_SIZE")],
                   "correction": "gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads -> doubled; WRITE_SIZE exact",
                   "hbm_bytes_per_launch_measured": hbm, "units_per_launch_measured": units, "hbm_bytes_per_unit": hbm / units,
                   "algorithmic_bytes_per_unit": 4 + 95232.0 / 16384 + 32.0 / 256,
                   "algorithmic_bytes_model": "4 B output + the pair's 95,232 B weight image staged once per 16,384-row superblock + 32 B of X per row shared by the launch's 256 pairs",
                   "SQ_INSTS_MFMA": fwd.get((k, "SQ_INSTS_MFMA")), "SQ_INSTS_VALU": fwd.get((k, "SQ_INSTS_VALU")),
                   "SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE": fwd.get((k, "GRBM_GUI_ACTIVE"))},
                  open(os.path.join(prof, f"{tag}_pmc_forward.json"), "w"), indent=1)
    trn = pmc(os.path.join(src, "pmc_train"), "train_")
    if trn:
        write_pmc(trn, os.path.join(prof, f"{tag}_pmc_train.csv"))
    red = pmc(os.path.join(src, "pmc_red"), "reduce_update4_kernel<true>")
    if red:
        write_pmc(red, os.path.join(prof, f"{tag}_pmc_reduce.csv"))
        k = next(k for (k, c) in red if c == "FETCH_SIZE")
        units = 262_144 * 8192
        hbm = (2 * red[(k, "FETCH_SIZE")] + red[(k, "WRITE_SIZE")]) * 1024.0
        json.dump({"kernel": k, "launch": "262,144 rows x 8,192 pairs, 128 bins (a row block of the bench)",
                   "FETCH_SIZE_KB": red[(k, "FETCH_SIZE")], "WRITE_SIZE_KB": red[(k, "WRITE_SIZE")],
                   "correction": "gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads -> doubled; WRITE_SIZE exact",
                   "hbm_bytes_per_launch_measured": hbm, "units_per_launch_measured": units, "hbm_bytes_per_unit": hbm / units,
                   "algorithmic_bytes_per_unit": 4 + 2.0 * (16 + 4 + 256) * 262_144 / units},
                  open(os.path.join(prof, f"{tag}_pmc_reduce.json"), "w"), indent=1)
    loss = {}
    for n in (36_500_000, 4_562_500):
        lp = pmc(os.path.join(src, f"pmc_loss_{n}"), "fused_")
        if not lp:
            continue
        per = {}
        for kern in ("fused_stats_kernel", "fused_forward_kernel", "fused_backward_kernel"):
            k = next((k for (k, c) in lp if kern in k and c == "FETCH_SIZE"), None)
            if k is None:
                continue
            fetch, write = lp[(k, "FETCH_SIZE")] * 1024.0, lp.get((k, "WRITE_SIZE"), 0.0) * 1024.0
            per[kern] = {"FETCH_SIZE_B_per_element": fetch / n, "fetched_B_per_element_corrected": 2 * fetch / n,
                         "WRITE_SIZE_B_per_element": write / n, "hbm_side_B_per_element": (2 * fetch + write) / n}
        loss[str(n)] = {"passes": per, "sum_B_per_element": sum(v["hbm_side_B_per_element"] for v in per.values()),
                        "algorithmic_B_per_element": 20, "moved_by_three_passes_B_per_element": 28,
                        "note": "FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced reads); the counters sit on the L2's "
                                "memory side, so Infinity-Cache hits are counted: a pass served from the 256 MiB cache still shows its bytes"}
    if loss:
        json.dump(loss, open(os.path.join(prof, f"{tag}_pmc_loss.json"), "w"), indent=1)
    print("digest written to profiles/ with tag", tag)


if __name__ == "__main__":
    main()

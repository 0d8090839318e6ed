#!/usr/bin/env python3
"""Condense gpurun_out/prof_r2 (tools/profile_r2.sh, run on the GPU box) into small summaries
under gpurun_out/prof_r2/summary/, which are then committed under profiles/ as round2_*."""
import csv
import glob
import json
import os
import shutil
import statistics
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", os.environ.get("IPS_PROF_DIR", "prof_r2"))
dst = os.path.join(src, "summary")
os.makedirs(dst, exist_ok=True)
tag = os.environ.get("IPS_PROF_TAG", "round2")


def one(pattern):
    m = glob.glob(os.path.join(src, pattern))
    return max(m, key=os.path.getmtime) if m else None


def rows_of(pattern):
    p = one(pattern)
    return list(csv.DictReader(open(p))) if p else []


def short(name):
    return name.split("(")[0].replace("void ", "")


def last_json(path):
    lines = [l for l in open(path).read().strip().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


# ---- 1. headline ---------------------------------------------------------------------------
shutil.copy(one("trace/*/*_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
trace = rows_of("trace/*/*_kernel_trace.csv")
scan = [r for r in trace if "fle_scan_kernel" in r["Kernel_Name"]]
gmax = max(int(r["Grid_Size_X"]) for r in scan)
big = [r for r in scan if int(r["Grid_Size_X"]) == gmax]
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in big]
bench_line = last_json(os.path.join(src, "bench_trace.json"))
plain_line = last_json(os.path.join(src, "bench_plain.json"))
with open(os.path.join(dst, f"{tag}_scan_kernel.md"), "w") as f:
    f.write(f"# {tag}: dominant kernel of bench.py under rocprofv3 --kernel-trace --stats\n\n")
    f.write("Command (on the MI355X box): `rocprofv3 --kernel-trace --stats --output-format csv -- "
            "python3 bench.py --steps 20 --warmup 3 --no-cpu --no-extra` (tools/profile_r2.sh).\n\n")
    f.write(f"Kernel: `{short(big[0]['Kernel_Name'])}`, grid {big[0]['Grid_Size_X']} threads, workgroup "
            f"{big[0]['Workgroup_Size_X']}, LDS {big[0]['LDS_Block_Size']} B/block, VGPR {big[0]['VGPR_Count']}, "
            f"SGPR {big[0]['SGPR_Count']}, scratch {big[0]['Scratch_Size']}.\n\n")
    f.write(f"| dispatches | avg us | min us | max us | avg of the 20 timed | avg of dispatches 15..28 (settled clock) |\n|---|---|---|---|---|---|\n")
    settled = dur[15:28] if len(dur) >= 28 else dur[-8:]
    f.write(f"| {len(dur)} | {statistics.mean(dur)/1e3:.1f} | {min(dur)/1e3:.1f} | {max(dur)/1e3:.1f} | "
            f"{statistics.mean(dur[3:23])/1e3:.1f} | {statistics.mean(settled)/1e3:.1f} |\n\n")
    ab = bench_line["roofline"]["algorithmic_bytes_per_launch"]
    rb = bench_line["roofline"]["read_bytes_per_launch"]
    t20 = statistics.mean(dur[3:23]) * 1e-9
    f.write(f"Algorithmic bytes per launch F = {ab} (reads {rb}): {ab/t20/1e9:.0f} GB/s = {ab/t20/8e12*100:.1f} % of 8 TB/s over the "
            f"20 timed dispatches, reads alone {rb/t20/1e9:.0f} GB/s = {rb/t20/8e12*100:.1f} %; settled: "
            f"{ab/(statistics.mean(settled)*1e-9)/1e9:.0f} GB/s.\n\n")
    f.write(f"bench.py's own HIP-event figure in the same run: kernel_ms_avg = {bench_line['roofline']['kernel_ms_avg']} ms; "
            f"un-profiled run: {plain_line['roofline']['kernel_ms_avg']} ms (frac {plain_line['roofline']['frac']}, "
            f"read-only {plain_line['roofline']['frac_read_only']}).\n\n")
    f.write("Per-dispatch durations (us): " + ", ".join(f"{d/1e3:.0f}" for d in dur) + "\n")
json.dump(plain_line, open(os.path.join(dst, f"{tag}_bench.json"), "w"), indent=1)


def largest(rows, substr, key="Grid_Size"):
    sc = [r for r in rows if substr in r["Kernel_Name"]]
    if not sc:
        return []
    g = max(int(r[key]) for r in sc)
    return [r for r in sc if int(r[key]) == g]


fetch = rows_of("fetch/*/*_counter_collection.csv")
write = rows_of("write/*/*_counter_collection.csv")
f_kib = statistics.mean(float(r["Counter_Value"]) for r in largest(fetch, "fle_scan_kernel"))
w_kib = statistics.mean(float(r["Counter_Value"]) for r in largest(write, "fle_scan_kernel"))
rows = bench_line["config"]["rows_per_gpu"]
bw = bench_line["config"]["bit_width"]
traffic = {
    "rows": rows, "bit_width": bw,
    "FETCH_SIZE_KiB_raw": round(f_kib, 1), "WRITE_SIZE_KiB_raw": round(w_kib, 1),
    "correction": "gfx950: FETCH_SIZE reports half of a wide coalesced streaming read (x2); "
                  "WRITE_SIZE exact for 16-byte streaming stores; both in KiB (x1024)",
    "hbm_read_bytes_per_launch": int(f_kib * 1024 * 2),
    "hbm_write_bytes_per_launch": int(w_kib * 1024),
    "hbm_bytes_per_launch": int(f_kib * 1024 * 2 + w_kib * 1024),
    "algorithmic_bytes_per_launch": bench_line["roofline"]["algorithmic_bytes_per_launch"],
    "source": f"profiles/{tag}_pmc_traffic.json from rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE "
              "(separate passes) over python3 bench.py --steps 5 --warmup 2 --no-cpu --no-extra",
}
json.dump(traffic, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)

sq = rows_of("sq/*/*_counter_collection.csv")
acc = defaultdict(list)
for r in largest(sq, "fle_scan_kernel"):
    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
tiles = (rows + 2047) // 2048
with open(os.path.join(dst, f"{tag}_sq_counters.md"), "w") as f:
    f.write(f"# {tag}: SQ counters of fle_scan_kernel<{bw},0,0>, {rows} rows ({tiles} sub-tiles)\n\n")
    f.write("| counter | per dispatch | per 2048-row sub-tile |\n|---|---|---|\n")
    for k in sorted(acc):
        v = statistics.mean(acc[k])
        f.write(f"| {k} | {v:.0f} | {v / tiles:.1f} |\n")

# ---- 2. the tour ---------------------------------------------------------------------------
p = one("tour_trace/*/*_kernel_stats.csv")
if p:
    shutil.copy(p, os.path.join(dst, f"{tag}_tour_kernel_stats.csv"))
tour = [json.loads(l) for l in open(os.path.join(src, "tour_plain.jsonl")) if l.startswith("{")]
tour_prof = {}
for l in open(os.path.join(src, "tour_trace.jsonl")):
    if l.startswith("{"):
        d = json.loads(l)
        tour_prof[d["op"]] = d
ttrace = rows_of("tour_trace/*/*_kernel_trace.csv")
tfetch = rows_of("tour_fetch/*/*_counter_collection.csv")
twrite = rows_of("tour_write/*/*_counter_collection.csv")
tsq = rows_of("tour_sq/*/*_counter_collection.csv")
with open(os.path.join(dst, f"{tag}_kernel_tour.md"), "w") as f:
    f.write(f"# {tag}: every kernel family under rocprofv3 (tools/kernel_tour.py, tools/profile_r2.sh)\n\n")
    f.write("2^28 rows (Q6: 600,037,902), inputs resident in HBM.  `us (trace)` = average duration of the op's "
            "dominant kernel in `rocprofv3 --kernel-trace` (largest grid of the matching name); `us (events)` = "
            "HIP-event time of the whole entry point in an un-profiled run (several launches for program / "
            "dictionary encode / nullable leaf).  GB/s = algorithmic bytes / us (events).  HBM MB = "
            "FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE of that kernel from separate --pmc passes.\n\n")
    f.write("| op | dominant kernel | dispatches | us (trace) | us (events) | algorithmic MB | GB/s | % of 8 TB/s | HBM MB (PMC) | VALU / LDS instr per 2048 rows | LDS conflict share |\n")
    f.write("|---|---|---|---|---|---|---|---|---|---|---|\n")
    fix = {"fle_pred w=32 BETWEEN": "fle_pred32_early_kernel<32, true>", "fle_pred w=16 BETWEEN": "fle_pred_w_kernel<16, 1>",
           "fle_pred w=8 BETWEEN": "fle_pred_w_kernel<8, 1>", "fle_pred w=16 LT": "fle_pred_w_kernel<16, 0>",
           "fle_pred w=8 LT": "fle_pred_w_kernel<8, 0>", "fle_pred w=32 LT": "fle_pred32_early_kernel<32, false>",
           "nullable leaf w=12, 10% NULL (counts + leaf)": "fle_leaf_kernel<12, 0>", "bitmap_expand (root 50%)": "expand_kernel<0, 0>",
           "Q6 conjunction, 3 columns, one-pass chain (1 launch)": "fle_chain_w_kernel<6, 16>",
           "Q6 conjunction, 3 columns, per-operand plan (3 launches)": "fle_pred_w_kernel<12, 1>"}
    for t in tour:
        t["kernel"] = fix.get(t["op"], t["kernel"])
        ks = largest(ttrace, t["kernel"], "Grid_Size_X")
        name = short(ks[0]["Kernel_Name"]) if ks else "?"
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ks]
        fk = [float(r["Counter_Value"]) for r in largest(tfetch, t["kernel"]) if r["Counter_Name"] == "FETCH_SIZE"]
        wk = [float(r["Counter_Value"]) for r in largest(twrite, t["kernel"]) if r["Counter_Name"] == "WRITE_SIZE"]
        hbm = (statistics.mean(fk) * 2048 + statistics.mean(wk) * 1024) / 1e6 if fk and wk else None
        sqv = defaultdict(list)
        for r in largest(tsq, t["kernel"]):
            sqv[r["Counter_Name"]].append(float(r["Counter_Value"]))
        n_tiles = (600037902 if t["op"].startswith("Q6") else (1 << 28)) / 2048
        valu = statistics.mean(sqv["SQ_INSTS_VALU"]) / n_tiles if sqv.get("SQ_INSTS_VALU") else None
        ldsi = statistics.mean(sqv["SQ_INSTS_LDS"]) / n_tiles if sqv.get("SQ_INSTS_LDS") else None
        conf = (statistics.mean(sqv["SQ_LDS_BANK_CONFLICT"]) / max(statistics.mean(sqv["SQ_LDS_IDX_ACTIVE"]), 1)
                if sqv.get("SQ_LDS_IDX_ACTIVE") else None)
        gbs = t["bytes"] / (t["us_event"] * 1e-6) / 1e9
        f.write(f"| {t['op']} | `{name}` | {len(d)} | {statistics.mean(d)/1e3:.1f} | {t['us_event']} | {t['bytes']/1e6:.0f} | "
                f"{gbs:.0f} | {gbs/80:.1f} | {hbm and round(hbm)} | "
                f"{valu and round(valu)} / {ldsi and round(ldsi, 1)} | {conf is not None and round(conf, 2)} |\n" if d else
                f"| {t['op']} | ? | 0 | - | {t['us_event']} | {t['bytes']/1e6:.0f} | {gbs:.0f} | {gbs/80:.1f} | - | - | - |\n")
print("summaries:", sorted(os.listdir(dst)))

# ---- 3. BASELINE configs[1..4] with their sweeps (tools/configs_bench.py) ------------------------
cb = os.path.join(src, "configs_bench.jsonl")
if os.path.exists(cb):
    recs = [json.loads(l) for l in open(cb) if l.startswith("{")]
    with open(os.path.join(dst, f"{tag}_configs_bench.md"), "w") as f:
        f.write(f"# {tag}: BASELINE configs[1..4] and their selectivity sweeps (tools/configs_bench.py, HIP-event timing, "
                "inputs resident in HBM; `check` = popcount / selected-row count against a torch model of the predicate)\n\n")
        f.write("| config | rows | algorithmic MB | us (median) | us (min) | GB/s | % of 8 TB/s | G rows/s | check |\n|---|---|---|---|---|---|---|---|---|\n")
        for r in recs:
            f.write(f"| {r['config']} | {r['rows']} | {r['algorithmic_bytes']/1e6:.0f} | {r['us_med']} | {r['us_min']} | "
                    f"{r['GBps_med']:.0f} | {r['GBps_med']/80:.1f} | {r['Grows_per_s_med']} | {r['check']} |\n")

#!/usr/bin/env python3
"""Condense gpurun_out/prof_rN (written by tools/profile.sh on the GPU box) into the committed
summaries under profiles/ and into profiles/traffic.json, which bench.py reads for
roofline.traffic.  usage: python tools/make_profiles.py [round_number]"""
import csv
import glob
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = int(sys.argv[1]) if len(sys.argv) > 1 else 1
src = os.path.join(ROOT, "gpurun_out", f"prof_r{rnd}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
tag = f"round{rnd}"


def one(pattern):
    # gpurun merges new outputs next to older ones: take the newest match
    return max(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)


def largest_scan(rows, grid_key):
    sc = [r for r in rows if "fle_scan_kernel" in r["Kernel_Name"]]
    g = max(int(r[grid_key]) for r in sc)
    return [r for r in sc if int(r[grid_key]) == g]


# 1. rocprofv3 --kernel-trace --stats summary of `python3 bench.py --steps 20 --warmup 3`
shutil.copy(one("trace/runc/*_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
trace = list(csv.DictReader(open(one("trace/runc/*_kernel_trace.csv"))))
big = largest_scan(trace, "Grid_Size_X")
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in big]
bench_line = json.loads(open(os.path.join(src, "bench_trace.json")).read().strip().splitlines()[-1])
plain_line = json.loads(open(os.path.join(src, "bench_plain.json")).read().strip().splitlines()[-1])
with open(os.path.join(dst, f"{tag}_scan_kernel.md"), "w") as f:
    f.write(f"# {tag}: dominant kernel of bench.py under rocprofv3 --kernel-trace --stats\n\n")
    f.write("Command (on the MI355X box): `rocprofv3 --kernel-trace --stats --output-format csv -- "
            "python3 bench.py --steps 20 --warmup 3 --no-cpu` (tools/profile.sh).\n\n")
    f.write(f"Kernel: `{big[0]['Kernel_Name'].split('(')[0]}`, grid {big[0]['Grid_Size_X']} threads, "
            f"workgroup {big[0]['Workgroup_Size_X']}, LDS {big[0]['LDS_Block_Size']} B/block, "
            f"VGPR {big[0]['VGPR_Count']}, SGPR {big[0]['SGPR_Count']}, scratch {big[0]['Scratch_Size']}.\n\n")
    f.write(f"The kernel_stats.csv row of this kernel also averages in the 2^20-row latency launches "
            f"(grid {min(int(r['Grid_Size_X']) for r in trace if 'fle_scan_kernel' in r['Kernel_Name'])}); "
            f"the 2^28-row dispatches alone ({len(dur)} = 3 warm-up + 20 timed):\n\n")
    f.write(f"| dispatches | avg us | min us | max us | avg of the 20 timed |\n|---|---|---|---|---|\n")
    f.write(f"| {len(dur)} | {statistics.mean(dur)/1e3:.1f} | {min(dur)/1e3:.1f} | {max(dur)/1e3:.1f} | "
            f"{statistics.mean(dur[-20:])/1e3:.1f} |\n\n")
    f.write(f"bench.py's own HIP-event figure in the same run: kernel_ms_avg = "
            f"{bench_line['roofline']['kernel_ms_avg']} ms, shortest of 5 individually bracketed launches after the region = {bench_line['roofline']['kernel_ms_min_of_5_after_region']} ms; "
            f"un-profiled run: {plain_line['roofline']['kernel_ms_avg']} / {plain_line['roofline']['kernel_ms_min_of_5_after_region']} ms.\n\n")
    f.write("Per-dispatch durations (us): " + ", ".join(f"{d/1e3:.0f}" for d in dur) + "\n")
shutil.copy(os.path.join(src, "bench_plain.json"), os.path.join(dst, f"{tag}_bench.json"))

# 2. HBM traffic: separate PMC passes, MI355X_MICROARCH.md "HBM" corrections
fetch = list(csv.DictReader(open(one("fetch/runc/*_counter_collection.csv"))))
write = list(csv.DictReader(open(one("write/runc/*_counter_collection.csv"))))
f_kib = statistics.mean(float(r["Counter_Value"]) for r in largest_scan(fetch, "Grid_Size"))
w_kib = statistics.mean(float(r["Counter_Value"]) for r in largest_scan(write, "Grid_Size"))
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
              "(separate passes) over python3 bench.py --steps 5 --warmup 2 --no-cpu",
}
json.dump(traffic, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)

# 3. SQ counters of the dominant kernel
sq = list(csv.DictReader(open(one("sq/runc/*_counter_collection.csv"))))
bigsq = largest_scan(sq, "Grid_Size")
acc = {}
for r in bigsq:
    acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
tiles = (rows + 2047) // 2048
with open(os.path.join(dst, f"{tag}_sq_counters.md"), "w") as f:
    f.write(f"# {tag}: SQ counters of fle_scan_kernel<{bw},0,0>, {rows} rows ({tiles} sub-tiles)\n\n")
    f.write("| counter | per dispatch | per 2048-row sub-tile |\n|---|---|---|\n")
    for k in sorted(acc):
        v = statistics.mean(acc[k])
        f.write(f"| {k} | {v:.0f} | {v / tiles:.1f} |\n")
print(json.dumps(traffic, indent=1))


# 5. BASELINE configs[2..4] (tools/configs_bench.py writes gpurun_out/configs_bench.json)
cfg = os.path.join(ROOT, "gpurun_out", "configs_bench.json")
if os.path.exists(cfg):
    rows = json.load(open(cfg))
    with open(os.path.join(dst, f"{tag}_configs_bench.md"), "w") as f:
        f.write(f"# {tag}: BASELINE.json configs[2..4] on one MI355X (tools/configs_bench.py)\n\n")
        f.write("Kernel time = median of 10 launches (HIP events), inputs resident in HBM. `check` = result "
                "equals an independent torch/numpy evaluation of the same predicate. `ips_eval_program` plans "
                "the tree per operand (stand-alone predicate kernels writing / AND-ing / OR-ing into a bitmap, "
                "both leaves of a BETWEEN in one pass) unless the row says otherwise.\n\n")
        f.write("| config | rows | algorithmic MB | median us | GB/s | % of 8 TB/s | Grows/s | check |\n")
        f.write("|---|---|---|---|---|---|---|---|\n")
        for r in rows:
            f.write(f"| {r['config']} | {r['rows']} | {r['algorithmic_bytes'] / 1e6:.0f} | {r['us_med']} | "
                    f"{r['GBps_med']} | {r['GBps_med'] / 80:.1f} | {r['Grows_per_s_med']} | {r['check']} |\n")

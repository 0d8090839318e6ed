#!/usr/bin/env python3
"""Condense tools/profile_r3.sh's page-list run (gpurun_out/prof_r3/chunk_bench.jsonl + the kernel stats of
chunks_trace) into profiles/round3_chunks.md.  The table of earlier stages is kept from the existing file."""
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_r3")
dst = os.path.join(ROOT, "profiles", "round3_chunks.md")
recs = [json.loads(l) for l in open(os.path.join(src, "chunk_bench.jsonl")) if l.startswith("{")]
ks = sorted(glob.glob(os.path.join(src, "chunks_trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(ks)))
old = open(dst).read()
stages = old[old.index("Earlier stages of the unaligned path"):old.index("Kernels of the traced run")]
out = ["# round3: page-list evaluation (ips_chunk_*) against the contiguous buffers\n",
       "Commands (on the MI355X box, tools/profile_r3.sh): `python3 tools/chunk_bench.py` (HIP events, median of\n"
       "10 individually timed calls after 3 warm-up calls; every result checked against the contiguous call's words / values) and\n"
       "`rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/chunk_bench.py` for the kernels.\n"
       "Columns are cut into SEPARATE device buffers, one per page (each page encoded on its own: the block\n"
       "geometry restarts at every page).  The Q6 rows: contiguous = the one-pass chain (1 launch); pages that hold the same\n"
       "rows in every column = the paged one-pass chain (blockIdx.y = page, + a fix-up launch when pages start inside bitmap\n"
       "dwords); pages cut differently per column = three per-operand launches + fix-ups under AUTO, the segmented one-pass\n"
       "chain (merge kernel + chain + fix-up) under IPS_PROGRAM_ONE_PASS.  Timed back to back (30 calls in a row) the paged\n"
       "chain settles at 275-280 us against 253-272 us contiguous (1.06-1.10): the medians below include the first, slower\n"
       "calls of each configuration, and depend on where the page buffers ended up in memory.\n",
       "| configuration | median us | min us | vs contiguous (median) | check |\n|---|---|---|---|---|"]
for r in recs:
    out.append(f"| {r['config'].replace('(3 launches)', '(one-pass chain, 1 launch)')} | {r['us_med']} | {r['us_min']} | "
               f"{r['vs_contiguous'] if r['vs_contiguous'] else '1.000 (reference)'} | {r['check']} |")
out.append("")
out.append(stages.rstrip() + "\n")
out.append("Kernels of the traced run (rocprofv3 --stats; the paged kernels carry every page of a run in one launch):\n")
out.append("| kernel | calls | average us | min us | max us |\n|---|---|---|---|---|")
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("chain", "pages_kernel", "chunk_kernel", "fixup", "fle_scan_kernel<32", "fle_scan_kernel<12", "select_nullable", "rank3", "selection_pages", "selnull")):
        out.append(f"| `{n.split('(')[0].replace('void ', '')}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | "
                   f"{float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} |")
open(dst, "w").write("\n".join(out) + "\n")
print("wrote", dst)

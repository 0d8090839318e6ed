#!/usr/bin/env python3
"""bench.py -- headline benchmark: fused FLE int32 (w=32) decode + LT predicate @10% selectivity.

One "step" = one pass of the hot path (ips_fle_scan: predicate on the encoded bit-planes ->
selection bitmap, selected rows decoded and written per 2048-row batch) over one batch of
synthetic column chunks already resident in HBM; with N > 1 each rank scans its own row stripe and
the step ends with an RCCL all-gather of the per-stripe bitmaps over xGMI.

Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement" for the byte accounting.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

METRIC = "decoded+filtered rows/sec and HBM GB/s vs roofline, int32 FLE @10% sel"
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=1 << 28,
                    help="rows per GPU per step (256 column chunks of 2^20 rows)")
    ap.add_argument("--bw", type=int, default=32)
    ap.add_argument("--sel", type=float, default=0.10)
    ap.add_argument("--cpu-rows", type=int, default=1 << 26, help="cpu_baseline sample rows")
    ap.add_argument("--no-cpu", action="store_true")
    return ap.parse_args()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch N>1 with "
                  "torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # the exchange step runs whenever there is more than one rank (IPS_BENCH_GATHER=1 forces it
    # under a 1-rank torchrun launch, to rehearse the code path on a single-GPU box)
    gather = world > 1 or (os.environ.get("IPS_BENCH_GATHER") == "1" and "RANK" in os.environ)
    if gather:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)  # nccl IS RCCL on ROCm

    ips = entry.load_package()
    capi = ips.capi
    capi.lib()  # fail loudly if the HIP library is missing: there is no fallback
    n, bw = args.rows, args.bw
    c = ips.synth.lt_constant(bw, args.sel)
    seed = ips.synth.SEED_HEADLINE + rank * n  # rank r holds rows [r*n, (r+1)*n) of the column

    # ---- synthetic column chunk(s), generated and FLE-encoded on the GPU ----------------------
    vals = capi.synth_u32(seed, n, bw, device=dev)
    enc = capi.fle_encode(vals, bw)
    torch.cuda.synchronize()
    # spot parity vs the oracle on the first 2^20 rows (config 1) before timing anything
    check = None
    O = None
    if rank == 0:
        O = entry.load_oracle()
        n1 = min(n, 1 << 20)
        host_vals = ips.synth.column_u32(seed, n1, bw)
        assert np.array_equal(vals[:n1].cpu().numpy().view(np.uint32), host_vals)
        enc1 = O.fle_encode(host_vals, bw)
        assert np.array_equal(enc[:len(enc1)].cpu().numpy().view(np.uint64), enc1)
    del vals
    torch.cuda.empty_cache()

    outputs = capi.alloc_scan_outputs(n, dev)
    words = (n + 63) // 64
    stream = torch.cuda.current_stream()
    # Exchange: the scan of step i+1 overlaps the all-gather of step i's bitmap (RCCL runs on its
    # own stream), so the local bitmap and the gathered bitmap are double-buffered.
    local_bm = [outputs[0], torch.empty_like(outputs[0])] if gather else [outputs[0]]
    full_bm = [torch.empty(words * world, dtype=torch.int64, device=dev) for _ in range(2)] \
        if gather else None
    pending = [None]

    def step(i):
        k = i & 1 if gather else 0
        outs = (local_bm[k], outputs[1], outputs[2])
        bitmap, bvals, counts = capi.fle_scan(enc, n, bw, capi.OP_LT, c, outputs=outs)
        if gather:
            work = dist.all_gather_into_tensor(full_bm[k], local_bm[k][:words], async_op=True)
            if pending[0] is not None:
                pending[0].wait()  # stream-side wait: buffer k^1 is free before step i+1 reuses it
            pending[0] = work
        return bitmap, bvals, counts

    def drain():
        if pending[0] is not None:
            pending[0].wait()
            pending[0] = None
        torch.cuda.synchronize()

    for i in range(args.warmup):
        bitmap, bvals, counts = step(i)
    if args.warmup == 0:
        bitmap, bvals, counts = step(0)
    drain()
    n_sel = int(counts.to(torch.int64).sum().item())
    if rank == 0:
        n1 = min(n, 1 << 20)
        bm_ref = O.fle_pred(enc1, n1, bw, O.OP_LT, c)
        got = bitmap[:len(bm_ref)].cpu().numpy().view(np.uint64)
        ok_bm = bool(np.array_equal(got, bm_ref))
        sel_ref = O.fle_select(enc1, n1, bw, bm_ref)
        nb1 = capi.n_batches(n1)
        cnt_h = counts[:nb1].cpu().numpy()
        bv_h = bvals[:nb1 * capi.BATCH_ROWS].cpu().numpy().view(np.uint32)
        got_sel = np.concatenate([bv_h[b * 2048: b * 2048 + cnt_h[b]] for b in range(nb1)])
        ok_sel = bool(np.array_equal(got_sel, sel_ref))
        ok_cnt = capi.bitmap_count(bitmap, n) == n_sel
        check = {"first_2^20_rows_bitmap_bit_exact": ok_bm, "selected_values_bit_exact": ok_sel,
                 "popcount_equals_batch_counts": bool(ok_cnt)}
        assert ok_bm and ok_sel and ok_cnt, check

    # ---- timed region: exactly K steps, barrier + synchronize on both sides ------------------
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    if gather:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    region = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    region[0].record(stream)
    for i in range(args.steps):
        k = i & 1 if gather else 0
        if gather:  # the stream also waits for gathers here: bracket every launch on its own
            ev[i][0].record(stream)
        capi.fle_scan(enc, n, bw, capi.OP_LT, c, outputs=(local_bm[k], outputs[1], outputs[2]))
        if gather:
            ev[i][1].record(stream)
        if gather:
            work = dist.all_gather_into_tensor(full_bm[k], local_bm[k][:words], async_op=True)
            if pending[0] is not None:
                pending[0].wait()
            pending[0] = work
    region[1].record(stream)
    drain()
    if gather:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if gather:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # average launch duration over the timed region, HIP events on the launch stream: one pair
    # around the K back-to-back launches (N = 1: nothing else is on the stream), per-launch pairs
    # when the stream also carries the waits for the all-gathers.  The shortest launch comes from
    # a few individually bracketed launches after the region.
    if gather:
        kern_ms = sorted(a.elapsed_time(b) for a, b in ev)
        kern_avg_ms = sum(kern_ms) / len(kern_ms)
    else:
        kern_avg_ms = region[0].elapsed_time(region[1]) / args.steps
        single = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                  for _ in range(5)]
        for a, b in single:
            a.record(stream)
            capi.fle_scan(enc, n, bw, capi.OP_LT, c, outputs=(local_bm[0], outputs[1], outputs[2]))
            b.record(stream)
        torch.cuda.synchronize()
        kern_ms = sorted(a.elapsed_time(b) for a, b in single)

    if gather:  # bit-identity: slice r of every gathered bitmap equals rank r's local bitmap
        for k in range(min(2, args.steps)):
            mine = full_bm[k][rank * words:(rank + 1) * words]
            assert torch.equal(mine, local_bm[k][:words]), "gathered bitmap differs from local"
        if world > 1:  # and every rank holds the same gathered words
            chk = full_bm[0].sum().reshape(1).clone()
            lo, hi = chk.clone(), chk.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            assert int(lo.item()) == int(hi.item()), "ranks disagree on the gathered bitmap"

    # ---- 1M-row single-chunk latency (config 1/2 literally: launch-bound) ---------------------
    n1 = 1 << 20
    lat_us = None
    graph_us = None
    if n >= n1:
        out1 = capi.alloc_scan_outputs(n1, dev)
        for _ in range(5):
            capi.fle_scan(enc, n1, bw, capi.OP_LT, c, outputs=out1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(reps):
            capi.fle_scan(enc, n1, bw, capi.OP_LT, c, outputs=out1)
        e1.record(stream)
        torch.cuda.synchronize()
        lat_us = e0.elapsed_time(e1) * 1000.0 / reps
        # the same 50 launches captured once into a hipGraph and replayed (the C-ABI launch path
        # makes no allocation / synchronisation, so it is capturable)
        graph_us = None
        try:
            side = torch.cuda.Stream()
            side.wait_stream(stream)
            with torch.cuda.stream(side):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    for _ in range(reps):
                        capi.fle_scan(enc, n1, bw, capi.OP_LT, c, outputs=out1)
                g.replay()
                torch.cuda.synchronize()
                g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                g0.record(side)
                for _ in range(5):
                    g.replay()
                g1.record(side)
                torch.cuda.synchronize()
                graph_us = g0.elapsed_time(g1) * 1000.0 / (5 * reps)
            stream.wait_stream(side)
        except Exception as ex:  # capture unsupported on this stack: report eager only
            graph_us = None
            print(f"bench.py: hipGraph capture skipped: {ex}", file=sys.stderr)

    # ---- the same column as 256 SEPARATE 2^20-row page buffers, ips_fle_scan_pages -----------
    pages_info = None
    if rank == 0 and n >= n1 and n % n1 == 0 and not gather:
        try:
            n_pages = n // n1
            wpp = n1 // 64 * bw
            pages = [(enc[p * wpp:(p + 1) * wpp].clone(), n1, capi.alloc_scan_outputs(n1, dev))
                     for p in range(n_pages)]
            plist = capi.make_page_list(pages)
            for _ in range(3):
                capi.fle_scan_pages(plist, bw, capi.OP_LT, c)
            pe = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                  for _ in range(10)]
            torch.cuda.synchronize()
            for a, b in pe:
                a.record(stream)
                capi.fle_scan_pages(plist, bw, capi.OP_LT, c)
                b.record(stream)
            torch.cuda.synchronize()
            ptimes = sorted(a.elapsed_time(b) for a, b in pe)
            same = all(torch.equal(pages[p][2][0][:n1 // 64], local_bm[0][p * n1 // 64:(p + 1) * n1 // 64])
                       for p in (0, n_pages // 2, n_pages - 1))
            pages_info = {"pages": n_pages, "rows_per_page": n1,
                          "us_per_step_median": round(ptimes[len(ptimes) // 2] * 1e3, 1),
                          "launches_per_step": (n_pages + 63) // 64, "bitmaps_equal_contiguous_run": same}
            del pages, plist
        except Exception as ex:  # dev information only
            print(f"bench.py: separate-pages leg skipped: {ex}", file=sys.stderr)

    if rank != 0:
        if gather:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (fle_scan_kernel<32, predicate>) --------------------
    blocks = (n + 63) // 64
    algo_bytes = 8 * bw * blocks + 8 * blocks + 4 * n_sel  # SURVEY 8(d) "F"
    achieved = algo_bytes / (kern_avg_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("rows") == n and tj.get("bit_width") == bw:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "kernel": f"ips::fle_scan_kernel<{bw},0,0>",
                "algorithmic_bytes_per_launch": algo_bytes,
                "kernel_ms_avg": round(kern_avg_ms, 4), "kernel_ms_min_of_5_after_region": round(kern_ms[0], 4)}

    # ---- CPU baseline: the oracle port on this box's host cores, bounded sample ---------------
    cpu = None
    if not args.no_cpu and world == 1:
        nc = min(args.cpu_rows, n)
        enc_h = enc[: (nc // 64) * bw].cpu().numpy().view(np.uint64)
        # thread counts to try: what the container may really use (affinity / cgroup quota) and
        # every online CPU; the better one is reported with its own thread count
        cands = sorted({O.hw_threads(), min(os.cpu_count() or 1, 256)})
        best, best_threads = {}, {}
        for threads in cands:
            for mode in (0, 1):
                O.bench_fused(enc_h, nc, bw, O.OP_LT, c, threads, mode)  # warm-up
                ts = []
                for _ in range(5):
                    t = time.perf_counter()
                    O.bench_fused(enc_h, nc, bw, O.OP_LT, c, threads, mode)
                    ts.append(time.perf_counter() - t)
                rate = nc / min(ts)
                if rate > best.get(mode, 0.0):
                    best[mode], best_threads[mode] = rate, threads
        top = max(best, key=best.get)
        cpu = {"value": round(best[top], 1), "unit": "rows/s", "cores": best_threads[top],
               "kind": "port",
               "sample": (f"first {nc} rows of the same column, same LT constant; oracle C port "
                          f"(scalar uint64 predicate as fle-encoding.h:8012-8066 + SWAR block unpack"
                          f"{', avx2 clones' if O.has_avx2() else ''}), one stripe per thread, best of 5, "
                          f"thread counts tried {cands}; one call per stripe {best[0]:.3e} rows/s "
                          f"({best_threads[0]} threads), reference-shaped 1024-row batches "
                          f"{best[1]:.3e} rows/s ({best_threads[1]} threads)"),
               "note": "the reference itself is unbuildable in this image (Boost/Impala headers absent), hence kind=port"}

    total_rows = n * world
    out = {
        "metric": METRIC, "value": round(total_rows * args.steps / elapsed, 1), "unit": "rows/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed * 1e3 / args.steps, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {
            "workload": ("BASELINE configs[1]: int32 FLE (w=32) column, single LT predicate @10% "
                         "selectivity, fused decode+LT kernel -> bitmap + selected values; "
                         f"{n >> 20} column chunks of 2^20 rows per GPU resident in HBM per step "
                         "(the 2^20-row single-chunk launch is latency-bound, see extra.latency)"),
            "rows_per_gpu": n, "bit_width": bw, "predicate": f"LT {c}",
            "selectivity": round(n_sel / n, 5), "batch_rows": capi.BATCH_ROWS,
            "parallelism": (f"{world} row stripes, RCCL all-gather of bitmap words per step "
                            "(gather of step i overlaps the scan of step i+1)"
                            if gather else "single GPU"),
        },
        "roofline": roofline, "cpu_baseline": cpu,
        "extra": {"check": check, "selected_rows": n_sel,
                  "latency": {"rows": n1, "us_per_launch_back_to_back": lat_us and round(lat_us, 2),
                              "us_per_launch_hipgraph_replay": graph_us and round(graph_us, 2)},
                  "separate_pages": pages_info,
                  "device": capi.device_info()[0]},
    }
    if gather:
        # the exchange next to the scan: what the stripes alone sustain, and what one step moves
        out["extra"]["exchange"] = {
            "scan_only_rows_per_s": round(total_rows / (kern_avg_ms * 1e-3), 1),
            "bitmap_bytes_sent_per_rank_per_step": words * 8,
            "bitmap_bytes_received_per_rank_per_step": words * 8 * (world - 1),
            "note": "value includes the all-gather (overlapped with the next step's scan); "
                    "scan_only = total rows / rank 0's average scan-kernel time in the same run"}
    print(json.dumps(out))
    if gather:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

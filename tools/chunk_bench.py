#!/usr/bin/env python3
"""Page-list evaluation against the contiguous buffer: the Q6 conjunction, the
dictionary IN scan + gather and the headline scan over SEPARATE page buffers per column
(ips_chunk_*), pages of 2^20 rows (aligned) and of 2^20 - 37 rows (every page starts at an odd bit
of the bitmap: the shifted / atomic-merge path)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

ips = entry.load_package()
capi = ips.capi
dev = torch.device("cuda")


def timeit(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return ts[len(ts) // 2], ts[0]


def page_cuts(n, rows):
    out, left = [], n
    while left > 0:
        m = min(rows, left)
        out.append(m)
        left -= m
    return out


SLAB = os.environ.get("IPS_PAGES_SLAB") == "1"


def make_chunk(codes, w, cuts):
    """separate device buffers, one per page (each encoded on its own: block geometry restarts).
    IPS_PAGES_SLAB=1: the page buffers are carved out of ONE allocation per column (256-byte aligned), as a scanner
    that reads a column chunk into one slab holds them -- many small allocations of the caching allocator cost the
    narrow kernels TLB reach."""
    pages, pos = [], 0
    encs = []
    for m in cuts:
        encs.append(capi.fle_encode(codes[pos:pos + m].clone(), w))
        pos += m
    if SLAB:
        offs, total = [], 0
        for e in encs:
            offs.append(total)
            total += (e.numel() * 8 + 255) // 256 * 256 // 8
        slab = torch.empty(total + 32, dtype=torch.int64, device=dev)
        views = []
        for e, o in zip(encs, offs):
            v = slab[o:o + e.numel()]
            v.copy_(e)
            views.append(v)
        encs = views
        KEEP.append(slab)
    for e, m in zip(encs, cuts):
        pages.append((e, m, w))
    return capi.Chunk(pages)


KEEP = []


out = []


def report(name, us_med, us_min, base_us, ok, **kw):
    d = dict(config=name, us_med=round(us_med, 1), us_min=round(us_min, 1), vs_contiguous=round(us_med / base_us, 3) if base_us else None, check=bool(ok), **kw)
    out.append(d)
    print(json.dumps(d), flush=True)


def main():
    q6 = ips.q6
    n = int(os.environ.get("IPS_Q6_ROWS", str(q6.ROWS)))
    codes = [q6.codes_gpu(capi, c, n) for c in range(3)]
    encs = [capi.fle_encode(codes[c], q6.COLUMNS[c][3]) for c in range(3)]
    nodes, cols = q6.program(capi, encs)
    bm = torch.empty((n + 63) // 64, dtype=torch.int64, device=dev)
    t_c, tmin = timeit(lambda: capi.eval_program(nodes, cols, n, bitmap=bm))
    report("Q6 conjunction, contiguous buffers (3 launches)", t_c, tmin, None, True, rows=n)
    ref = bm.clone()
    del encs, cols
    for label, rows in (("2^20-row pages", 1 << 20), ("pages of 2^20 - 37 rows (unaligned)", (1 << 20) - 37),
                        ("column pages of different sizes (2^20, 2^20 - 37, 700001 rows)", None)):
        sizes = [rows] * 3 if rows else [1 << 20, (1 << 20) - 37, 700001]
        chunks = [make_chunk(codes[c], q6.COLUMNS[c][3], page_cuts(n, sizes[c])) for c in range(3)]
        bm2 = torch.empty_like(bm)
        t, tmin = timeit(lambda: capi.eval_program_chunks(nodes, chunks, bitmap=bm2))
        report(f"Q6 conjunction over page lists, {label}", t, tmin, t_c, torch.equal(bm2, ref),
               pages_per_column=[len(page_cuts(n, s)) for s in sizes])
        if rows is None:  # the segmented one-pass chain (IPS_PROGRAM_ONE_PASS; AUTO keeps the per-operand launches)
            capi.set_program_strategy(capi.PROGRAM_ONE_PASS)
            t, tmin = timeit(lambda: capi.eval_program_chunks(nodes, chunks, bitmap=bm2))
            capi.set_program_strategy(capi.PROGRAM_AUTO)
            report(f"Q6 conjunction over page lists, {label}: segmented one-pass chain (ONE_PASS)", t, tmin, t_c,
                   torch.equal(bm2, ref))
        for ch in chunks:
            ch.close()
        del chunks
    del codes, bm, ref
    torch.cuda.empty_cache()

    # dictionary IN scan + gather, D = 4096 (w = 12), K = 16, 2^28 rows
    n = 1 << 28
    rng = np.random.default_rng(4)
    D, K, bw = 4096, 16, 12
    dict_vals = np.sort(rng.choice(np.arange(-2 ** 30, 2 ** 30, 7), D, replace=False)).astype(np.int32)
    codes = ((capi.synth_u32(ips.synth.SEED_DICT, n, 32).to(torch.int64) & 0xFFFFFFFF) % D).to(torch.int32)
    enc = capi.fle_encode(codes, bw)
    dd = capi.Dict(dict_vals.view(np.uint8), capi.T_INT32)
    present = rng.choice(D, K // 2, replace=False)
    lits = np.concatenate([dict_vals[present], dict_vals[present] + 1]).astype(np.int32)
    res = {}

    def f():
        res["r"] = dd.scan(enc, n, bw, capi.OP_IN, lits)
    t_c, tmin = timeit(f, reps=6)
    report("dictionary IN K=16 scan + gather D=4096, contiguous", t_c, tmin, None, True, rows=n)
    ref_bm = res["r"][0].clone()
    ref_dense = capi.batches_compact(res["r"][1], res["r"][2], n)
    del res, enc
    for label, rows in (("2^20-row pages", 1 << 20), ("pages of 2^20 - 37 rows (unaligned)", (1 << 20) - 37)):
        chunk = make_chunk(codes, bw, page_cuts(n, rows))
        outs = chunk.alloc_outputs()
        t, tmin = timeit(lambda: chunk.dict_scan(dd, capi.OP_IN, lits, outputs=outs), reps=6)
        ok = torch.equal(outs[0][:ref_bm.numel()], ref_bm) and torch.equal(chunk.compact(outs[1], outs[2][:chunk.n_batches]), ref_dense)
        report(f"dictionary IN K=16 scan + gather over a page list, {label}", t, tmin, t_c, ok, pages=len(page_cuts(n, rows)))
        chunk.close()
        del outs
    dd.close()
    del codes, ref_bm, ref_dense
    torch.cuda.empty_cache()

    # headline: w = 32 LT @10 %
    bw = 32
    c = ips.synth.lt_constant(bw)
    vals = capi.synth_u32(ips.synth.SEED_HEADLINE, n, bw)
    enc = capi.fle_encode(vals, bw)
    outs0 = capi.alloc_scan_outputs(n, dev)
    t_c, tmin = timeit(lambda: capi.fle_scan(enc, n, bw, capi.OP_LT, c, outputs=outs0))
    report("headline w=32 LT @10% fused scan, contiguous", t_c, tmin, None, True, rows=n)
    ref_bm = outs0[0].clone()
    del enc
    for label, rows in (("2^20-row pages", 1 << 20), ("pages of 2^20 - 37 rows (unaligned)", (1 << 20) - 37)):
        chunk = make_chunk(vals, bw, page_cuts(n, rows))
        outs = chunk.alloc_outputs()
        t, tmin = timeit(lambda: chunk.fle_scan(capi.OP_LT, c, outputs=outs))
        ok = torch.equal(outs[0][:ref_bm.numel()], ref_bm)
        report(f"headline fused scan over a page list, {label}", t, tmin, t_c, ok, pages=len(page_cuts(n, rows)))
        chunk.close()
        del outs
    json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "chunk_bench.json"), "w"), indent=1)


if __name__ == "__main__":
    main()

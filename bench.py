#!/usr/bin/env python3
"""bench.py -- headline benchmark: fused FLE int32 (w=32) decode + LT predicate @10% selectivity.

One "step" = one pass of the hot path (ips_fle_scan: predicate on the encoded bit-planes ->
selection bitmap, selected rows decoded and written per 2048-row batch) over one batch of
synthetic column chunks already resident in HBM.  With N > 1 every rank scans its own block-cyclic
row stripes chunk by chunk and the bitmap words of chunk i are all-gathered (RCCL over xGMI through
the C-ABI's ips_allgather_bitmap, on its own stream) while chunk i+1 is being scanned.

Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement" for the byte accounting.
"""
import argparse
import ctypes as C
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
N_CHUNKS = int(os.environ.get("IPS_BENCH_CHUNKS", "8"))  # exchange granularity inside one step (N > 1)


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
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary legs (configs, h2d, ...)")
    return ap.parse_args()


def ev():
    return torch.cuda.Event(enable_timing=True)


def time_launches(fn, reps=10, warm=2):
    """median / min seconds of individually bracketed launches on the current stream"""
    for _ in range(warm):
        fn()
    pairs = [(ev(), ev()) for _ in range(reps)]
    torch.cuda.synchronize()
    for a, b in pairs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e-3 for a, b in pairs)
    return ts[len(ts) // 2], ts[0]


def pack_mask(mask):
    n = mask.numel()
    pad = (-n) % 64
    if pad:
        mask = torch.cat([mask, torch.zeros(pad, dtype=torch.bool, device=mask.device)])
    w = mask.view(-1, 64).to(torch.int64)
    return (w << torch.arange(64, device=w.device, dtype=torch.int64)).sum(dim=1)


def rec(name, rows, byts, tmed, tmin, check, **kw):
    d = {"config": name, "rows": rows, "algorithmic_bytes": int(byts), "us_med": round(tmed * 1e6, 1),
         "us_min": round(tmin * 1e6, 1), "GBps": round(byts / tmed / 1e9, 1),
         "frac": round(byts / tmed / 1e9 / HBM_PEAK_GBS, 4), "rows_per_s": round(rows / tmed, 1),
         "check": check}
    d.update(kw)
    return d


# ---- secondary legs (N = 1, rank 0): the other BASELINE configs as driver-run numbers ----------
def leg_widths(ips, capi, dev, n):
    """configs[1] at the code widths real FLE_DICTIONARY pages have: fused scan, LT @10 %."""
    out = []
    for bw in (16, 12, 8):
        vals = capi.synth_u32(ips.synth.SEED_HEADLINE, n, bw, device=dev)
        enc = capi.fle_encode(vals, bw)
        c = ips.synth.lt_constant(bw)
        outs = capi.alloc_scan_outputs(n, dev)
        tmed, tmin = time_launches(lambda: capi.fle_scan(enc, n, bw, capi.OP_LT, c, outputs=outs))
        n_sel = int(outs[2].to(torch.int64).sum().item())
        ok = n_sel == int((vals < c).sum().item()) == capi.bitmap_count(outs[0], n)
        blocks = (n + 63) // 64
        out.append(rec(f"configs[1] FLE w={bw} fused scan LT @10%", n, 8 * bw * blocks + 8 * blocks + 4 * n_sel,
                       tmed, tmin, bool(ok), selectivity=round(n_sel / n, 4)))
        del vals, enc, outs
    return out


def leg_plain_twin(ips, capi, dev, n):
    """configs[0]/[1], the PLAIN twin: the headline column as little-endian int32 slots, LT with the
    reference's reversed operands (parquet-common.h:208-217) and with SQL's; predicate and fused scan."""
    out = []
    page = capi.synth_u32(ips.synth.SEED_HEADLINE, n, 32, device=dev)   # int32 slots (signed)
    c = int(ips.synth.lt_constant(32))
    W = (n + 63) // 64
    for sem, name in ((capi.SEM_REFERENCE, "REFERENCE (literal < x)"), (capi.SEM_SQL, "SQL (x < literal)")):
        exp = int(((page > c) if sem == capi.SEM_REFERENCE else (page < c)).sum().item())
        bm = torch.empty(W, dtype=torch.int64, device=dev)
        tmed, tmin = time_launches(lambda: capi.plain_pred(page, n, capi.T_INT32, capi.OP_LT, np.int32(c), sem, bitmap=bm))
        out.append(rec(f"configs[1] PLAIN int32 twin, LT {name}: predicate", n, 4 * n + 8 * W, tmed, tmin,
                       capi.bitmap_count(bm, n) == exp, selectivity=round(exp / n, 4)))
        res = {}

        def f():
            res["r"] = capi.plain_scan(page, n, capi.T_INT32, capi.OP_LT, np.int32(c), semantics=sem)
        tmed, tmin = time_launches(f, reps=6)
        bm2, bv, cnt = res["r"]
        n_sel = int(cnt.to(torch.int64).sum().item())
        out.append(rec(f"configs[1] PLAIN int32 twin, LT {name}: fused scan (bitmap + selected slots)", n,
                       4 * n + 8 * W + 4 * n_sel, tmed, tmin, bool(n_sel == exp and torch.equal(bm2, bm)),
                       selectivity=round(n_sel / n, 4)))
        del res, bm2, bv, cnt
    return out


def leg_config2(ips, capi, dev, n):
    """configs[2]: int64 -- PLAIN 8 B/row and dictionary D = 4096 (w = 12); BETWEEN @10 %."""
    out = []
    W = (n + 63) // 64
    lo32 = capi.synth_u32(0x5EED0003, n, 32, device=dev).to(torch.int64) & 0xFFFFFFFF
    hi8 = capi.synth_u32(0x5EED1003, n, 8, device=dev).to(torch.int64)
    plain64 = (hi8 << 32) | lo32                         # values mod 2^40
    del lo32, hi8
    sel = 0.10
    lo, hi = int((0.5 - sel / 2) * (1 << 40)), int((0.5 + sel / 2) * (1 << 40))
    cols = [capi.plain_column(plain64, capi.T_INT64)]
    nodes = [capi.plain_leaf(0, capi.OP_GE, np.int64(lo), capi.T_INT64),
             capi.plain_leaf(0, capi.OP_LE, np.int64(hi), capi.T_INT64), capi.and_node()]
    bm = torch.empty(W, dtype=torch.int64, device=dev)
    tmed, tmin = time_launches(lambda: capi.eval_program(nodes, cols, n, bitmap=bm))
    exp = int(((plain64 >= lo) & (plain64 <= hi)).sum().item())
    out.append(rec("configs[2] PLAIN int64 BETWEEN @10% (one pass, And(Ge,Le))", n, 8 * n + n // 8, tmed, tmin,
                   capi.bitmap_count(bm, n) == exp, selectivity=round(exp / n, 4)))
    # the second form SURVEY 8d names: And(Gt a, Lt b) (simple-predicates.h:145-153)
    nodes = [capi.plain_leaf(0, capi.OP_GT, np.int64(lo - 1), capi.T_INT64),
             capi.plain_leaf(0, capi.OP_LT, np.int64(hi + 1), capi.T_INT64), capi.and_node()]
    bm_gl = torch.empty(W, dtype=torch.int64, device=dev)
    tmed, tmin = time_launches(lambda: capi.eval_program(nodes, cols, n, bitmap=bm_gl))
    out.append(rec("configs[2] PLAIN int64 And(Gt a, Lt b) @10% (one pass)", n, 8 * n + n // 8, tmed, tmin,
                   bool(torch.equal(bm_gl, bm)), selectivity=round(exp / n, 4)))
    del plain64, bm_gl
    D = 4096
    rng = np.random.default_rng(3)
    dict_vals = np.sort(rng.choice(np.arange(-2 ** 40, 2 ** 40, 2 ** 18), D, replace=False)).astype(np.int64)
    codes = ((capi.synth_u32(0x5EED0003, n, 32, device=dev).to(torch.int64) & 0xFFFFFFFF) % D).to(torch.int32)
    enc = capi.fle_encode(codes, 12)
    dd = capi.Dict(dict_vals.view(np.uint8), capi.T_INT64)
    lo, hi = dict_vals[int(0.45 * (D - 1))], dict_vals[int(0.55 * (D - 1))]
    _, op_lo, c_lo = dd.translate(capi.OP_GE, lo)
    _, op_hi, c_hi = dd.translate(capi.OP_LE, hi)
    nodes = [capi.leaf(0, op_lo, c_lo), capi.leaf(0, op_hi, c_hi), capi.and_node()]
    cols = [capi.fle_column(enc, 12)]
    tmed, tmin = time_launches(lambda: capi.eval_program(nodes, cols, n, bitmap=bm))
    lo_c, hi_c = int(np.searchsorted(dict_vals, lo)), int(np.searchsorted(dict_vals, hi, side="right"))
    exp = int(((codes >= lo_c) & (codes < hi_c)).sum().item())
    out.append(rec("configs[2] dictionary int64 D=4096 w=12 BETWEEN @10%", n, 12 * 8 * W + 8 * W + D * 8, tmed, tmin,
                   capi.bitmap_count(bm, n) == exp, selectivity=round(exp / n, 4)))
    # And(Gt a, Lt b) with a, b between dictionary entries: DictDecoder::Gt -> Ge(upper_bound),
    # ::Lt -> Lt(lower_bound) (dict-encoding.h:473-495); the same rows as the BETWEEN above
    _, op_a, c_a = dd.translate(capi.OP_GT, lo - 1)
    _, op_b, c_b = dd.translate(capi.OP_LT, hi + 1)
    nodes = [capi.leaf(0, op_a, c_a), capi.leaf(0, op_b, c_b), capi.and_node()]
    bm_gl = torch.empty(W, dtype=torch.int64, device=dev)
    tmed, tmin = time_launches(lambda: capi.eval_program(nodes, cols, n, bitmap=bm_gl))
    out.append(rec("configs[2] dictionary int64 D=4096 w=12 And(Gt a, Lt b) @10%", n, 12 * 8 * W + 8 * W + D * 8,
                   tmed, tmin, bool(torch.equal(bm_gl, bm)), selectivity=round(exp / n, 4)))
    dd.close()
    return out


def leg_config3(ips, capi, dev, n):
    """configs[3]: dictionary int32, IN list (half absent) + gather of the selected rows."""
    out = []
    W = (n + 63) // 64
    rng = np.random.default_rng(4)
    for D, K in ((256, 16), (4096, 16), (40000, 4)):
        bw = capi.dict_bit_width(D)
        dict_vals = np.sort(rng.choice(np.arange(-2 ** 30, 2 ** 30, 7), D, replace=False)).astype(np.int32)
        codes = ((capi.synth_u32(ips.synth.SEED_DICT, n, 32, device=dev).to(torch.int64) & 0xFFFFFFFF) % D).to(torch.int32)
        enc = capi.fle_encode(codes, bw)
        dd = capi.Dict(dict_vals.view(np.uint8), capi.T_INT32)
        present = rng.choice(D, K // 2, replace=False)
        lits = np.concatenate([dict_vals[present], dict_vals[present] + 1]).astype(np.int32)
        res = {}

        def f():
            res["r"] = dd.scan(enc, n, bw, capi.OP_IN, lits)
        tmed, tmin = time_launches(f, reps=6)
        bitmap, bvals, counts = res["r"]
        n_sel = int(counts.to(torch.int64).sum().item())
        exp = int(torch.isin(codes, torch.tensor(np.sort(present), device=dev, dtype=torch.int32)).sum().item())
        out.append(rec(f"configs[3] dictionary int32 D={D} w={bw} IN K={K} fused scan+gather", n,
                       bw * 8 * W + 8 * W + 4 * n_sel + D * 4, tmed, tmin, n_sel == exp,
                       selectivity=round(n_sel / n, 5)))
        dd.close()
        del codes, enc, res
    return out


def leg_nullable(capi, dev, n):
    """An OPTIONAL w=12 column, 10 % NULLs: the fused nullable leaf (def levels + data -> bitmap)."""
    nn = capi.synth_u32(0x5EED0D1, n, 32, device=dev)
    is_set = (nn.to(torch.int64) & 0xFFFFFFFF) >= int(0.1 * (1 << 32))
    del nn
    defs = capi.fle_encode(is_set.to(torch.int32), 1)
    k = int(is_set.sum().item())
    vals = capi.synth_u32(0x5EED0D2, k, 12, device=dev)
    enc = capi.fle_encode(vals, 12)
    n_data = ((k + 63) // 64) * 64
    ws = capi.nullable_workspace(n, dev)
    bm = torch.empty((n + 63) // 64, dtype=torch.int64, device=dev)
    tmed, tmin = time_launches(lambda: capi.fle_pred_nullable(defs, 1, 1, n, enc, n_data, 12, capi.OP_LT, 409,
                                                              bitmap=bm, workspace=ws))
    sel_rows = torch.zeros(n, dtype=torch.bool, device=dev)
    sel_rows[is_set] = vals < 409
    ok = capi.bitmap_count(bm, n) == int(sel_rows.sum().item())
    byts = (n + 63) // 64 * 8 * 2 + n_data // 64 * 12 * 8
    out = [rec("OPTIONAL FLE w=12 column, 10% NULL, LT @10%: nullable leaf (levels + data -> bitmap)", n, byts,
               tmed, tmin, bool(ok), non_null_rows=k)]
    # late materialisation of the same column for the rows the leaf selected (ReadValue(skip) over the
    # whole selection): dense values of the selected rows + one NOT-NULL flag per selected row
    import ctypes as C
    lib = capi.lib()
    lib.ips_select_nullable_workspace_bytes.restype = C.c_size_t
    ws2 = torch.empty(int(lib.ips_select_nullable_workspace_bytes(C.c_int64(n), C.c_int64(n_data), 4)) + 16,
                      dtype=torch.uint8, device=dev)
    n_sel = int(sel_rows.sum().item())
    dense = torch.empty(n_sel + 64, dtype=torch.int32, device=dev)
    flags = torch.empty((n + 63) // 64, dtype=torch.int64, device=dev)
    cnts = torch.zeros(3, dtype=torch.int64, device=dev)
    P = lambda t: C.c_void_p(t.data_ptr())
    stream = capi._stream(None)

    def mat():
        capi._ck(lib.ips_dict_select_nullable(None, P(defs), 1, 1, C.c_int64(n), P(enc), C.c_int64(n_data), 12,
                                              P(bm), P(dense), P(flags), P(cnts), P(ws2), stream))
    tmed, tmin = time_launches(mat)
    got = cnts.cpu().tolist()[:2]
    ok2 = (got == [n_sel, n_sel] and torch.equal(dense[:n_sel], vals[vals < 409].to(torch.int32))
           and capi.bitmap_count(flags, n_sel) == n_sel)
    byts2 = (n + 63) // 64 * 8 * 2 + n_data // 64 * 12 * 8 + 4 * n_sel + n_sel // 8
    out.append(rec("OPTIONAL FLE w=12 column, 10% NULL: late materialisation of the leaf's selection "
                   "(levels + selection + data -> dense values + NOT-NULL flags, 2 launches)", n, byts2, tmed, tmin,
                   bool(ok2), selected_rows=n_sel))
    t_contig = tmed
    # the same over the column as a LIST OF PAGES (separate buffers per page, levels and data blocks restart at
    # every page): ips_chunk_select_nullable, 4 launches whatever the number of pages
    for label, rows in (("256 pages of 2^20 rows", 1 << 20), ("257 pages of 2^20 - 37 rows", (1 << 20) - 37)):
        pages, pos, dpos = [], 0, 0
        while pos < n:
            m = min(rows, n - pos)
            s_p = is_set[pos:pos + m]
            k_p = int(s_p.sum().item())
            pages.append((capi.fle_encode(vals[dpos:dpos + k_p].clone(), 12) if k_p else None, m, 12,
                          capi.fle_encode(s_p.to(torch.int32), 1), k_p))
            pos += m
            dpos += k_p
        chunk = capi.Chunk(pages, max_def_level=1)
        lib.ips_chunk_select_nullable_workspace_bytes.restype = C.c_size_t
        lib.ips_chunk_select_nullable_workspace_bytes.argtypes = [C.c_void_p]
        ws3 = torch.empty(int(lib.ips_chunk_select_nullable_workspace_bytes(chunk.h)) + 16, dtype=torch.uint8, device=dev)
        dense_p = torch.empty(n_sel + 64, dtype=torch.int32, device=dev)
        flags_p = torch.empty((n + 63) // 64, dtype=torch.int64, device=dev)
        cnts_p = torch.zeros(3, dtype=torch.int64, device=dev)

        def mat_pages():
            capi._ck(lib.ips_chunk_select_nullable(chunk.h, None, P(bm), P(dense_p), P(flags_p), P(cnts_p), P(ws3), stream))
        tmed, tmin = time_launches(mat_pages)
        ok3 = (cnts_p.cpu().tolist() == [n_sel, n_sel, 0] and torch.equal(dense_p[:n_sel], dense[:n_sel])
               and torch.equal(flags_p[:(n_sel + 63) // 64], flags[:(n_sel + 63) // 64]))
        out.append(rec(f"OPTIONAL FLE w=12 column, 10% NULL: late materialisation over a page list, {label} (4 launches)",
                       n, byts2, tmed, tmin, bool(ok3), vs_contiguous=round(tmed / t_contig, 3)))
        chunk.close()
        del pages, ws3, dense_p, flags_p
    return out


def q6_piece(ips, capi, dev, row0, rows):
    codes = [ips.q6.codes_gpu(capi, c, rows, start=row0, device=dev) for c in range(3)]
    encs = [capi.fle_encode(codes[c], ips.q6.COLUMNS[c][3]) for c in range(3)]
    return codes, encs


def leg_q6_single(ips, capi, dev, O):
    """configs[4] on one GPU: the three-column conjunction over all 600,037,902 rows."""
    q6 = ips.q6
    n = q6.ROWS
    codes, encs = q6_piece(ips, capi, dev, 0, n)
    nodes, cols = q6.program(capi, encs)
    bm = torch.empty((n + 63) // 64, dtype=torch.int64, device=dev)
    tmed, tmin = time_launches(lambda: capi.eval_program(nodes, cols, n, bitmap=bm))
    mask = q6.truth(codes)
    ok = torch.equal(pack_mask(mask), bm)
    n1 = 1 << 20
    ops = {"GE": O.OP_GE, "LT": O.OP_LT}
    ref = None
    for col, op, k in q6.LEAVES:
        e = O.fle_encode(q6.codes_numpy(col, n1), q6.COLUMNS[col][3])
        leaf = O.fle_pred(e, n1, q6.COLUMNS[col][3], ops[op], k)
        ref = leaf if ref is None else (ref & leaf)
    ok_oracle = np.array_equal(bm[:n1 // 64].cpu().numpy().view(np.uint64), ref)
    return [rec("configs[4] TPC-H-Q6 shape: 3 dictionary columns (w=12,4,6) x 600,037,902 rows, conjunction, 1 GPU",
                n, q6.algorithmic_bytes(n), tmed, tmin, bool(ok and ok_oracle),
                selectivity=round(int(mask.sum().item()) / n, 5),
                check_detail="every bit vs torch on the raw codes; first 2^20 rows vs the oracle")]


def leg_page_lists(ips, capi, dev):
    """The page loop of the scanner inside the launches (ips_chunk_*): configs[4]'s conjunction and
    configs[3]'s IN scan + gather over SEPARATE page buffers per column, 2^20-row pages and pages whose
    ends differ between the columns (every boundary inside a bitmap word), against the contiguous call."""
    out = []
    q6 = ips.q6
    n = q6.ROWS

    def chunk_of(vals, w, rows):
        pages, pos = [], 0
        while pos < vals.numel():
            m = min(rows, vals.numel() - pos)
            pages.append((capi.fle_encode(vals[pos:pos + m].clone(), w), m, w))
            pos += m
        return capi.Chunk(pages)
    codes = [q6.codes_gpu(capi, c, n, device=dev) for c in range(3)]
    encs = [capi.fle_encode(codes[c], q6.COLUMNS[c][3]) for c in range(3)]
    nodes, cols = q6.program(capi, encs)
    ref = capi.eval_program(nodes, cols, n)
    t_c, _ = time_launches(lambda: capi.eval_program(nodes, cols, n, bitmap=ref), reps=20, warm=5)
    del encs, cols
    bm = torch.empty_like(ref)
    # pages that hold the same rows in every column: the one-pass chain, blockIdx.y = page (1 launch; with
    # page starts inside bitmap dwords + 1 fix-up launch); page ends that differ between the columns: 3
    # per-operand launches + their fix-ups
    for label, sizes, launches in (("573 pages of 2^20 rows per column", (1 << 20,) * 3, 1),
                                   ("573 pages of 2^20 - 37 rows per column (every page starts inside a bitmap dword)",
                                    ((1 << 20) - 37,) * 3, 2),
                                   ("pages of 2^20 / 2^20 - 37 / 700,001 rows (page ends differ between the columns)",
                                    (1 << 20, (1 << 20) - 37, 700001), 5)):
        chunks = [chunk_of(codes[c], q6.COLUMNS[c][3], sizes[c]) for c in range(3)]
        tmed, tmin = time_launches(lambda: capi.eval_program_chunks(nodes, chunks, bitmap=bm), reps=20, warm=5)
        out.append(rec(f"configs[4] Q6 conjunction over page lists, {label}", n, q6.algorithmic_bytes(n), tmed, tmin,
                       bool(torch.equal(bm, ref)), vs_contiguous=round(tmed / t_c, 3),
                       launches_per_step=launches))
        for ch in chunks:
            ch.close()
        del chunks
    del codes, ref, bm
    torch.cuda.empty_cache()
    n = 1 << 28
    W = n // 64
    rng = np.random.default_rng(4)
    D, K, bw = 4096, 16, 12
    dict_vals = np.sort(rng.choice(np.arange(-2 ** 30, 2 ** 30, 7), D, replace=False)).astype(np.int32)
    codes = ((capi.synth_u32(ips.synth.SEED_DICT, n, 32, device=dev).to(torch.int64) & 0xFFFFFFFF) % D).to(torch.int32)
    enc = capi.fle_encode(codes, bw)
    dd = capi.Dict(dict_vals.view(np.uint8), capi.T_INT32)
    present = rng.choice(D, K // 2, replace=False)
    lits = np.concatenate([dict_vals[present], dict_vals[present] + 1]).astype(np.int32)
    res = {}

    def f():
        res["r"] = dd.scan(enc, n, bw, capi.OP_IN, lits)
    t_c, _ = time_launches(f, reps=6)
    ref_bm = res["r"][0].clone()
    n_sel = int(res["r"][2].to(torch.int64).sum().item())
    del res, enc
    for label, rows in (("256 pages of 2^20 rows", 1 << 20), ("257 pages of 2^20 - 37 rows", (1 << 20) - 37)):
        chunk = chunk_of(codes, bw, rows)
        outs = chunk.alloc_outputs()
        tmed, tmin = time_launches(lambda: chunk.dict_scan(dd, capi.OP_IN, lits, outputs=outs), reps=6)
        ok = torch.equal(outs[0][:W], ref_bm) and int(outs[2][:chunk.n_batches].to(torch.int64).sum().item()) == n_sel
        out.append(rec(f"configs[3] dictionary int32 D=4096 w=12 IN K=16 scan+gather over a page list, {label}", n,
                       bw * 8 * W + 8 * W + 4 * n_sel + D * 4, tmed, tmin, bool(ok), vs_contiguous=round(tmed / t_c, 3)))
        chunk.close()
        del outs
    dd.close()
    return out


def leg_h2d(capi, dev, enc, n, bw, c, outputs, hbm_rows_per_s):
    """The same column handed over as HOST memory: pinned buffer, 16 chunks, two device staging
    buffers; the copy of chunk i+1 overlaps the scan of chunk i.  PCIe-inclusive rate, never `value`."""
    n_chunks = 16
    rows_c = n // n_chunks
    if rows_c % 2048 or rows_c == 0:
        return None
    words_c = rows_c // 64 * bw
    host = torch.empty(enc.numel(), dtype=torch.int64, pin_memory=True)
    host.copy_(enc)
    torch.cuda.synchronize()
    stage = [torch.empty(words_c, dtype=torch.int64, device=dev) for _ in range(2)]
    copy_s, scan_s = torch.cuda.Stream(), torch.cuda.Stream()
    bitmap, bvals, counts = outputs
    bitmap2 = torch.zeros_like(bitmap)

    def run():
        copied = [None] * n_chunks
        scanned = [None] * n_chunks
        for i in range(n_chunks):
            with torch.cuda.stream(copy_s):
                if i >= 2:
                    copy_s.wait_event(scanned[i - 2])      # staging buffer i%2 is free again
                stage[i % 2].copy_(host[i * words_c:(i + 1) * words_c], non_blocking=True)
                copied[i] = copy_s.record_event()
            with torch.cuda.stream(scan_s):
                scan_s.wait_event(copied[i])
                outs = (bitmap2[i * rows_c // 64:], bvals[i * rows_c:], counts[i * rows_c // 2048:])
                capi.fle_scan(stage[i % 2], rows_c, bw, capi.OP_LT, c, outputs=outs, stream=scan_s)
                scanned[i] = scan_s.record_event()
    run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    t = min(ts)
    same = torch.equal(bitmap2[:n // 64], bitmap[:n // 64])
    return {"rows": n, "chunks": n_chunks, "host_bytes": enc.numel() * 8, "ms": round(t * 1e3, 3),
            "h2d_GBps": round(enc.numel() * 8 / t / 1e9, 1), "rows_per_s_pcie_inclusive": round(n / t, 1),
            "vs_hbm_resident": round(n / t / hbm_rows_per_s, 4), "bitmap_equals_resident_run": bool(same),
            "note": "pinned host buffer, 2 device staging buffers, copy of chunk i+1 overlaps the scan of chunk i"}


def cpu_baseline(O, ips, enc, n, bw, c, cpu_rows):
    nc = min(cpu_rows, n)
    enc_h = enc[: (nc // 64) * bw].cpu().numpy().view(np.uint64)
    buffers = O.bench_buffers(nc)          # allocated and touched once, outside every timed run
    usable = O.hw_threads()                # min(online CPUs, affinity mask, cgroup quota)
    cands = sorted({1, usable})
    res = {}
    for threads in cands:
        for mode in (0, 1):
            sec, cnt = O.bench_fused_best(enc_h, nc, bw, O.OP_LT, c, threads, mode, buffers, reps=5)
            res[(threads, mode)] = nc / sec
    best_key = max(res, key=res.get)
    one = max(res[(1, 0)], res[(1, 1)])
    return {"value": round(res[best_key], 1), "unit": "rows/s", "cores": usable, "threads": best_key[0],
            "kind": "port",
            "sample": (f"first {nc} rows of the same column, same LT constant; oracle C port (scalar uint64 "
                       f"predicate as fle-encoding.h:8012-8066 + block unpack"
                       f"{' with AVX2' if O.has_avx2() else ' (SWAR)'}), one 1024-row-aligned stripe per thread, "
                       f"buffers pre-touched, best of 5 timed inside C after a warm-up pass"),
            "single_thread_rows_per_s": round(one, 1),
            "modes": {f"{t} threads, {'1024-row batches' if m else 'one call per stripe'}": round(v, 1)
                      for (t, m), v in res.items()},
            "host_cpus_online": os.cpu_count(),
            "note": ("the reference itself is unbuildable in this image (Boost/Impala headers absent), hence "
                     "kind=port; cores = CPUs this container may use (affinity / cgroup quota)")}


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
    ips = entry.load_package()
    capi = ips.capi
    capi.lib()  # fail loudly if the HIP library is missing: there is no fallback
    comm = None
    if gather:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)  # bootstrap, barriers, max-over-ranks only
        # the data-path exchange goes through the C-ABI (what a C++ host calls): RCCL communicator
        # created from an id that rank 0 makes and the bootstrap distributes
        idt = torch.zeros(capi.COMM_ID_BYTES, dtype=torch.uint8, device=dev)
        # RCCL prints a version banner on stdout (through the C library's buffer, whenever it likes):
        # fd 1 points at stderr until the ONE JSON line is due
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(capi.comm_unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, 0)
        comm = capi.Comm(bytes(idt.cpu().numpy().tobytes()), world, rank)
        torch.cuda.synchronize()

    n, bw = args.rows, args.bw
    c = ips.synth.lt_constant(bw, args.sel)
    n_chunks = N_CHUNKS if gather else 1
    rows_c = n // n_chunks
    assert rows_c % 2048 == 0 and rows_c * n_chunks == n, "--rows must be a multiple of 16384"
    words_c = rows_c // 64
    enc_words_c = words_c * bw

    # ---- synthetic column, generated and FLE-encoded on the GPU --------------------------------
    # N = 1: rows [0, n).  N > 1: block-cyclic stripes (sharding.cyclic_pieces): this rank's chunk i
    # is piece i*world + rank of the global column of n*world rows, so that the all-gather of chunk
    # i lands at words [i*world*words_c, (i+1)*world*words_c) of the global bitmap, natural order.
    O = entry.load_oracle() if rank == 0 else None
    enc = torch.empty(n // 64 * bw, dtype=torch.int64, device=dev)
    for i in range(n_chunks):
        g_row0 = (i * world + rank) * rows_c
        vals = capi.synth_u32(ips.synth.SEED_HEADLINE + g_row0, rows_c, bw, device=dev)
        capi.fle_encode(vals, bw, out=enc[i * enc_words_c:(i + 1) * enc_words_c])
        if rank == 0 and i == 0:
            # spot parity vs the oracle on the first 2^20 rows (config 1) before timing anything
            n1 = min(rows_c, 1 << 20)
            host_vals = ips.synth.column_u32(ips.synth.SEED_HEADLINE, n1, bw)
            assert np.array_equal(vals[:n1].cpu().numpy().view(np.uint32), host_vals)
            enc1 = O.fle_encode(host_vals, bw)
            torch.cuda.synchronize()
            assert np.array_equal(enc[:len(enc1)].cpu().numpy().view(np.uint64), enc1)
        del vals
    torch.cuda.synchronize()
    torch.cuda.empty_cache()

    outputs = capi.alloc_scan_outputs(n, dev)
    words = n // 64
    stream = torch.cuda.current_stream()
    comm_stream = torch.cuda.Stream() if gather else None  # (the Q6 leg's exchange stream)
    # N > 1: one C call per step (ips_fle_scan_allgather) issues the chunk scans on `stream` and the
    # all-gather of every finished chunk on the communicator's own stream; the next step first
    # waits (stream-side, ips_comm_join) until the previous step's gathers have read the buffers.
    local_bm = outputs[0]
    full_bm = torch.empty(words * world, dtype=torch.int64, device=dev) if gather else None

    def step(s, bracket=None):
        if gather:
            comm.join(stream)
        if bracket:
            bracket[0].record(stream)
        if gather:
            comm.fle_scan_allgather(enc, n, bw, capi.OP_LT, c, n_chunks, local_bm, outputs[1], outputs[2],
                                    full_bm, stream=stream)
        else:
            capi.fle_scan(enc, n, bw, capi.OP_LT, c, outputs=outputs)
        if bracket:
            bracket[1].record(stream)

    def drain():
        if gather:
            comm.join(stream)
        torch.cuda.synchronize()

    for s in range(max(args.warmup, 1)):
        step(s)
    drain()
    bitmap, bvals, counts = local_bm, outputs[1], outputs[2]
    n_sel = int(counts.to(torch.int64).sum().item())
    check = None
    if rank == 0:
        n1 = min(rows_c, 1 << 20)
        bm_ref = O.fle_pred(enc1, n1, bw, O.OP_LT, c)
        got = bitmap[:len(bm_ref)].cpu().numpy().view(np.uint64)
        ok_bm = bool(np.array_equal(got, bm_ref))
        sel_ref = O.fle_select(enc1, n1, bw, bm_ref)
        nb1 = capi.n_batches(n1)
        cnt_h = counts[:nb1].cpu().numpy()
        bv_h = bvals[:nb1 * capi.BATCH_ROWS].cpu().numpy().view(np.uint32)
        got_sel = np.concatenate([bv_h[b * 2048: b * 2048 + cnt_h[b]] for b in range(nb1)])
        ok_sel = bool(np.array_equal(got_sel, sel_ref))
        ok_cnt = capi.bitmap_count(bitmap[:words], n) == n_sel
        check = {"first_2^20_rows_bitmap_bit_exact": ok_bm, "selected_values_bit_exact": ok_sel,
                 "popcount_equals_batch_counts": bool(ok_cnt)}
        assert ok_bm and ok_sel and ok_cnt, check

    # ---- timed region: exactly K steps, barrier + synchronize on both sides --------------------
    brackets = [(ev(), ev()) for _ in range(args.steps)]
    if gather:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    region = (ev(), ev())
    region[0].record(stream)
    for s in range(args.steps):
        step(s, brackets[s] if gather else None)
    region[1].record(stream)
    drain()
    if gather:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if gather:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # average scan time per step, HIP events on the launch stream: one pair around the K
    # back-to-back launches (N = 1: nothing else is on the stream); with the exchange, one pair
    # around each step's chunk launches (the stream also waits for buffer reuse between steps).
    if gather:
        kern_ms = sorted(a.elapsed_time(b) for a, b in brackets)
        kern_avg_ms = sum(kern_ms) / len(kern_ms)
    else:
        kern_avg_ms = region[0].elapsed_time(region[1]) / args.steps
        single = [(ev(), ev()) for _ in range(5)]
        for a, b in single:
            a.record(stream)
            capi.fle_scan(enc, n, bw, capi.OP_LT, c, outputs=outputs)
            b.record(stream)
        torch.cuda.synchronize()
        kern_ms = sorted(a.elapsed_time(b) for a, b in single)

    if gather:  # bit-identity of the exchange
        comm.check()  # no waiter of a step gave up
        for i in range(n_chunks):  # piece (i, rank) of the gathered bitmap is this rank's chunk i
            g0 = (i * world + rank) * words_c
            assert torch.equal(full_bm[g0:g0 + words_c], local_bm[i * words_c:(i + 1) * words_c]), \
                "gathered bitmap differs from the local chunk"
        chk = full_bm.sum().reshape(1).clone()  # every rank holds the same gathered words
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert int(lo.item()) == int(hi.item()), "ranks disagree on the gathered bitmap"
        if rank == 0:  # natural order: words of global rows [0, 2^20) are rank 0's first words
            n1 = min(rows_c, 1 << 20)
            assert np.array_equal(full_bm[:n1 // 64].cpu().numpy().view(np.uint64), bm_ref)

    # ---- configs[4] sharded: the Q6 conjunction over block-cyclic stripes + chunked gather ------
    q6_sharded = None
    if gather and not args.no_extra:
        try:  # a secondary leg: its failure must not take the headline line with it
            q6_sharded = leg_q6_sharded(ips, capi, dev, dist, comm, comm_stream, world, rank)
        except Exception as ex:
            q6_sharded = {"config": "configs[4] TPC-H-Q6 shape sharded", "error": repr(ex), "check": False}
            print(f"bench.py: Q6 sharded leg failed on rank {rank}: {ex!r}", file=sys.stderr)

    extra = {}
    if rank == 0 and not gather and not args.no_extra:
        n1 = 1 << 20
        # ---- 1M-row single-chunk latency (config 1/2 literally: launch-bound) ----------------
        if n >= n1:
            out1 = capi.alloc_scan_outputs(n1, dev)
            for _ in range(5):
                capi.fle_scan(enc, n1, bw, capi.OP_LT, c, outputs=out1)
            e0, e1 = ev(), ev()
            reps = 50
            torch.cuda.synchronize()
            e0.record(stream)
            for _ in range(reps):
                capi.fle_scan(enc, n1, bw, capi.OP_LT, c, outputs=out1)
            e1.record(stream)
            torch.cuda.synchronize()
            lat_us = e0.elapsed_time(e1) * 1000.0 / reps
            graph_us = None
            try:  # the same 50 launches captured once into a hipGraph and replayed
                side = torch.cuda.Stream()
                side.wait_stream(stream)
                with torch.cuda.stream(side):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=side):
                        for _ in range(reps):
                            capi.fle_scan(enc, n1, bw, capi.OP_LT, c, outputs=out1)
                    g.replay()
                    torch.cuda.synchronize()
                    g0, g1 = ev(), ev()
                    g0.record(side)
                    for _ in range(5):
                        g.replay()
                    g1.record(side)
                    torch.cuda.synchronize()
                    graph_us = g0.elapsed_time(g1) * 1000.0 / (5 * reps)
                stream.wait_stream(side)
            except Exception as ex:  # capture unsupported on this stack: report eager only
                print(f"bench.py: hipGraph capture skipped: {ex}", file=sys.stderr)
            extra["latency"] = {"rows": n1, "us_per_launch_back_to_back": round(lat_us, 2),
                                "us_per_launch_hipgraph_replay": graph_us and round(graph_us, 2)}
        # ---- the same column as 256 SEPARATE 2^20-row page buffers, ips_fle_scan_pages --------
        if n >= n1 and n % n1 == 0:
            try:
                n_pages = n // n1
                wpp = n1 // 64 * bw
                pages = [(enc[p * wpp:(p + 1) * wpp].clone(), n1, capi.alloc_scan_outputs(n1, dev))
                         for p in range(n_pages)]
                plist = capi.make_page_list(pages)
                tmed, tmin = time_launches(lambda: capi.fle_scan_pages(plist, bw, capi.OP_LT, c), reps=10, warm=3)
                same = all(torch.equal(pages[p][2][0][:n1 // 64], outputs[0][p * n1 // 64:(p + 1) * n1 // 64])
                           for p in (0, n_pages // 2, n_pages - 1))
                extra["separate_pages"] = {"pages": n_pages, "rows_per_page": n1,
                                           "us_per_step_median": round(tmed * 1e6, 1),
                                           "launches_per_step": (n_pages + 63) // 64,
                                           "bitmaps_equal_contiguous_run": same}
                del pages, plist
            except Exception as ex:  # dev information only
                print(f"bench.py: separate-pages leg skipped: {ex}", file=sys.stderr)
        hbm_rate = n / (kern_avg_ms * 1e-3)
        try:
            extra["h2d"] = leg_h2d(capi, dev, enc, n, bw, c, outputs, hbm_rate)
        except Exception as ex:
            print(f"bench.py: h2d leg skipped: {ex}", file=sys.stderr)

    # ---- CPU baseline: the oracle port on this box's host cores, bounded sample ---------------
    cpu = None
    if rank == 0 and not args.no_cpu and world == 1:
        cpu = cpu_baseline(O, ips, enc, n, bw, c, args.cpu_rows)

    # ---- the other BASELINE configs on this GPU (driver-run numbers with their checks) -------
    if rank == 0 and not gather and not args.no_extra:
        del enc
        torch.cuda.empty_cache()
        configs = []
        for leg in (lambda: leg_widths(ips, capi, dev, n), lambda: leg_plain_twin(ips, capi, dev, n),
                    lambda: leg_config2(ips, capi, dev, n),
                    lambda: leg_config3(ips, capi, dev, n), lambda: leg_nullable(capi, dev, n),
                    lambda: leg_q6_single(ips, capi, dev, O), lambda: leg_page_lists(ips, capi, dev)):
            try:
                configs += leg()
            except Exception as ex:
                configs.append({"config": "leg failed", "error": repr(ex), "check": False})
            torch.cuda.empty_cache()
        extra["configs"] = configs

    if rank != 0:
        if gather:
            comm.close()
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (fle_scan_kernel<32, predicate>) --------------------
    blocks = (n + 63) // 64
    read_bytes = 8 * bw * blocks
    algo_bytes = read_bytes + 8 * blocks + 4 * n_sel  # SURVEY 8(d) "F"
    achieved = algo_bytes / (kern_avg_ms * 1e-3) / 1e9
    read_only = read_bytes / (kern_avg_ms * 1e-3) / 1e9
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("rows") == n and tj.get("bit_width") == bw:
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_src = "not measured in this run: " + tj.get("source", "profiles/traffic.json")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": traffic_src,
                "read_only_GBps": round(read_only, 1), "frac_read_only": round(read_only / HBM_PEAK_GBS, 4),
                "kernel": f"ips::fle_scan_kernel<{bw},0,0>",
                "algorithmic_bytes_per_launch": algo_bytes, "read_bytes_per_launch": read_bytes,
                "kernel_ms_avg": round(kern_avg_ms, 4),
                "kernel_ms_min_of_5_after_region" if not gather else "step_scan_ms_min": round(kern_ms[0], 4),
                "note": ("frac counts the bytes of SURVEY 8(d) F = encoded planes + bitmap + 4 B per selected "
                         "row; frac_read_only counts the encoded planes alone (north_star's HBM-read roofline)")}

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
            "parallelism": (f"{world} ranks, block-cyclic row stripes ({n_chunks} chunks of {rows_c} rows per rank "
                            "and step, one ips_fle_scan_allgather call); RCCL all-gather of chunk i on the "
                            "communicator's stream while chunk i+1 is scanned; gathered bitmap in natural row order"
                            if gather else "single GPU"),
        },
        "roofline": roofline, "cpu_baseline": cpu,
        "extra": dict({"check": check, "selected_rows": n_sel, "device": capi.device_info()[0]}, **extra),
    }
    if gather:
        # the exchange next to the scan: what the stripes alone sustain, and what one step moves
        out["extra"]["exchange"] = {
            "scan_only_rows_per_s": round(total_rows / (kern_avg_ms * 1e-3), 1),
            "chunks_per_step": n_chunks,
            "bitmap_bytes_sent_per_rank_per_step": words * 8,
            "bitmap_bytes_received_per_rank_per_step": words * 8 * (world - 1),
            "note": "value includes the all-gathers (chunk i's exchange overlaps the scan of chunk i+1 and of the "
                    "next step); scan_only = total rows / rank 0's average per-step scan time in the same run"}
        out["extra"]["q6_sharded"] = q6_sharded
    if gather:
        C.CDLL(None).fflush(None)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
    print(json.dumps(out), flush=True)
    if gather:
        os.dup2(2, 1)
        comm.close()
        dist.destroy_process_group()


def leg_q6_sharded(ips, capi, dev, dist, comm, comm_stream, world, rank):
    """configs[4] as north_star states it: the three Q6 columns over 600,037,902 rows sharded over
    the ranks (block-cyclic stripes, all columns cut alike), ips_eval_program per chunk, RCCL
    all-gather of the chunk's bitmap words overlapped with the next chunk (strong scaling)."""
    q6, sh = ips.q6, ips.sharding
    n = q6.ROWS
    piece_rows, pieces = sh.cyclic_pieces(n, world, rank, N_CHUNKS)
    pw = piece_rows // 64
    stream = torch.cuda.current_stream()
    # every rank holds N_CHUNKS pieces of piece_rows rows per column (pieces beyond the data are padded
    # with rows that select nothing: code 0 fails "shipdate >= 365"), each column as ONE ips_chunk whose
    # pages are the pieces; one C call per step: ips_eval_program_chunks_allgather
    data = []
    col_pages = [[], [], []]
    for row0, row1 in pieces:
        m = max(row1 - row0, 0)
        codes = [q6.codes_gpu(capi, c, m, start=row0, device=dev) if m else torch.zeros(0, dtype=torch.int32, device=dev)
                 for c in range(3)]
        data.append((m, codes))
        for c in range(3):
            padded = torch.zeros(piece_rows, dtype=torch.int32, device=dev)
            padded[:m] = codes[c]
            w = q6.COLUMNS[c][3]
            col_pages[c].append((capi.fle_encode(padded, w), piece_rows, w))
    chunks = [capi.Chunk(col_pages[c]) for c in range(3)]
    nodes, _ = q6.program(capi, [pg[0][0] for pg in col_pages])
    local = torch.zeros(N_CHUNKS * pw, dtype=torch.int64, device=dev)
    full = torch.empty(N_CHUNKS * world * pw, dtype=torch.int64, device=dev)

    def run():
        comm.eval_program_chunks_allgather(nodes, chunks, local, full, stream=stream)
        comm.join(stream)

    for _ in range(2):
        run()
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    t = float(el.item()) / reps
    # checks: this rank's pieces against torch on the raw codes; all ranks hold the same words
    ok = True
    for i, (m, codes) in enumerate(data):
        if m == 0:
            continue
        g0 = (i * world + rank) * pw
        exp = pack_mask(q6.truth(codes))
        ok = ok and torch.equal(full[g0:g0 + exp.numel()], exp)
    comm.check()
    chk = full[:(n + 63) // 64].sum().reshape(1).clone()
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    okt = torch.tensor([int(ok and int(lo.item()) == int(hi.item()))], device=dev)
    dist.all_reduce(okt, op=dist.ReduceOp.MIN)
    return {"config": "configs[4] TPC-H-Q6 shape sharded: 3 columns x 600,037,902 rows over the ranks + all-gather of bitmaps",
            "rows": n, "ranks": world, "chunks": N_CHUNKS, "piece_rows": piece_rows, "ms": round(t * 1e3, 4),
            "rows_per_s": round(n / t, 1), "algorithmic_bytes": q6.algorithmic_bytes(n),
            "GBps_aggregate": round(q6.algorithmic_bytes(n) / t / 1e9, 1), "scaling": "strong",
            "check": bool(okt.item()),
            "launches_per_step": "one one-pass chain launch over all pieces (ips_eval_program_chunks_allgather) + one "
                                 "waiter and one ncclAllGather per piece on the communicator's stream",
            "check_detail": "each rank's gathered pieces vs torch on the raw codes; all ranks hold identical words"}


if __name__ == "__main__":
    main()

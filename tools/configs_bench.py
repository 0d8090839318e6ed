#!/usr/bin/env python3
"""Secondary benchmark (dev tool, 1 GPU): BASELINE.json configs[2..4] -- kernel time, algorithmic
GB/s and a parity / property check for each.  The headline (configs[1]) is bench.py.
usage: python tools/configs_bench.py [--q6-rows N]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402


def timeit(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    torch.cuda.synchronize()
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[0] * 1e-3, ts[len(ts) // 2] * 1e-3


def codes_column(capi, seed, n, D):
    x = capi.synth_u32(seed, n, 32)
    return ((x.to(torch.int64) & 0xFFFFFFFF) % D).to(torch.int32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--q6-rows", type=int, default=600_037_902)
    ap.add_argument("--rows", type=int, default=1 << 28)
    args = ap.parse_args()
    ips = entry.load_package()
    capi = ips.capi
    O = entry.load_oracle()
    dev = torch.device("cuda")
    out = []

    def report(name, rows, byts, tmin, tmed, ok, extra=None):
        rec = {"config": name, "rows": rows, "algorithmic_bytes": int(byts), "us_min": round(tmin * 1e6, 1),
               "us_med": round(tmed * 1e6, 1), "GBps_med": round(byts / tmed / 1e9, 1),
               "Grows_per_s_med": round(rows / tmed / 1e9, 1), "check": ok}
        if extra:
            rec.update(extra)
        out.append(rec)
        print(json.dumps(rec), flush=True)

    n = args.rows
    W = (n + 63) // 64
    # ---- configs[1] beyond its quoted point: the fused FLE scan over the selectivity sweep ----------
    for bw in (32, 12):
        vals = capi.synth_u32(0x5EED0001, n, bw)
        enc = capi.fle_encode(vals, bw)
        outs = capi.alloc_scan_outputs(n, dev)
        for sel in (0.01, 0.10, 0.50, 1.0):
            c = min(int(sel * (1 << bw)), (1 << bw) - 1)
            op = capi.OP_LT if sel < 1.0 else capi.OP_LE
            f = lambda: capi.fle_scan(enc, n, bw, op, c, outputs=outs)
            tmin, tmed = timeit(f)
            nsel = int(outs[2].to(torch.int64).sum().item())
            exp = int(((vals.to(torch.int64) & 0xFFFFFFFF) < c).sum().item()) if sel < 1.0 else n
            report(f"configs[1] FLE w={bw} fused scan LT sel={sel}", n, W * 8 * (bw + 1) + 4 * nsel, tmin, tmed,
                   nsel == exp and capi.bitmap_count(outs[0], n) == exp, {"selectivity": round(nsel / n, 4)})
        del vals, enc, outs
    # ---- configs[2]: int64 -- PLAIN 8 B/row and dictionary D=4096 (w=12); BETWEEN + And(Gt,Lt) ---
    rng = np.random.default_rng(3)
    x = capi.synth_u32(0x5EED0003, n, 32).to(torch.int64) & 0xFFFFFFFF
    x2 = capi.synth_u32(0x5EED1003, n, 8).to(torch.int64)
    plain64 = ((x2 << 32) | x)                      # values mod 2^40, non-negative int64
    del x, x2
    for sel in (0.01, 0.10, 0.50, 1.0):
        lo = int((0.5 - sel / 2) * (1 << 40))
        hi = int((0.5 + sel / 2) * (1 << 40)) - (0 if sel < 1.0 else 1)
        cols = [capi.plain_column(plain64, capi.T_INT64)]
        nodes = [capi.plain_leaf(0, capi.OP_GE, np.int64(lo), capi.T_INT64),
                 capi.plain_leaf(0, capi.OP_LE, np.int64(hi), capi.T_INT64), capi.and_node()]
        bm = torch.empty(W, dtype=torch.int64, device=dev)
        f = lambda: capi.eval_program(nodes, cols, n, bitmap=bm)
        tmin, tmed = timeit(f)
        cnt = capi.bitmap_count(bm, n)
        exp = int(((plain64 >= lo) & (plain64 <= hi)).sum().item())
        report(f"configs[2] PLAIN int64 BETWEEN sel={sel}", n, 8 * n + n / 8, tmin, tmed, cnt == exp,
               {"selectivity": round(cnt / n, 4)})
        if sel in (0.01, 0.10):
            # the same predicate with the selected rows materialised: fused scan vs pred + select
            import ctypes as C
            lib = capi.lib()
            P_ = lambda t: C.c_void_p(t.data_ptr())
            St = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            bv = torch.empty(((n + 2047) // 2048) * 2048, dtype=torch.int64, device=dev)
            cn = torch.empty((n + 2047) // 2048, dtype=torch.int32, device=dev)
            lo_a, hi_a = np.array([lo], np.int64), np.array([hi], np.int64)
            fs = lambda: lib.ips_plain_scan(P_(plain64), C.c_int64(n), capi.T_INT64, capi.OP_GE,
                                            lo_a.ctypes.data_as(C.c_void_p), 1, capi.OP_LE,
                                            hi_a.ctypes.data_as(C.c_void_p), capi.SEM_SQL, P_(bm), P_(bv),
                                            P_(cn), St)
            tmin, tmed = timeit(fs)
            ok = int(cn.to(torch.int64).sum().item()) == exp and capi.bitmap_count(bm, n) == exp
            report(f"configs[2] PLAIN int64 BETWEEN sel={sel} fused scan (bitmap + selected slots)", n,
                   8 * n + n / 8 + 8 * exp, tmin, tmed, ok)

            def two_pass():
                capi.eval_program(nodes, cols, n, bitmap=bm)
                lib.ips_plain_select(P_(plain64), C.c_int64(n), capi.T_INT64, P_(bm), P_(bv), P_(cn), St)
            tmin, tmed = timeit(two_pass)
            report(f"configs[2] PLAIN int64 BETWEEN sel={sel} predicate, then select against the bitmap", n,
                   8 * n + n / 8 + 8 * exp, tmin, tmed, int(cn.to(torch.int64).sum().item()) == exp)
            del bv, cn
    del plain64
    D = 4096
    dict_vals = np.sort(rng.choice(np.arange(-2 ** 40, 2 ** 40, 2 ** 18), D, replace=False)).astype(np.int64)
    codes = codes_column(capi, 0x5EED0003, n, D)
    enc = capi.fle_encode(codes, 12)
    dd = capi.Dict(dict_vals.view(np.uint8), capi.T_INT64)
    for sel in (0.01, 0.10, 0.50, 1.0):
        lo = dict_vals[int((0.5 - sel / 2) * (D - 1))]
        hi = dict_vals[int((0.5 + sel / 2) * (D - 1))]
        _, op_lo, c_lo = dd.translate(capi.OP_GE, lo)
        _, op_hi, c_hi = dd.translate(capi.OP_LE, hi)
        k_lo, k_hi = dd.translate(capi.OP_GE, lo)[0], dd.translate(capi.OP_LE, hi)[0]
        cols = [capi.fle_column(enc, 12)]
        nodes = []
        nodes.append(capi.leaf(0, op_lo, c_lo) if k_lo == capi.XL_FLE else capi.leaf(0, capi.OP_GE, 0))
        nodes.append(capi.leaf(0, op_hi, c_hi) if k_hi == capi.XL_FLE else capi.leaf(0, capi.OP_GE, 0))
        nodes.append(capi.and_node())
        bm = torch.empty(W, dtype=torch.int64, device=dev)
        f = lambda: capi.eval_program(nodes, cols, n, bitmap=bm)
        tmin, tmed = timeit(f)
        cnt = capi.bitmap_count(bm, n)
        lo_c, hi_c = int(np.searchsorted(dict_vals, lo)), int(np.searchsorted(dict_vals, hi, side="right"))
        exp = int(((codes >= lo_c) & (codes < hi_c)).sum().item())
        report(f"configs[2] dict int64 D=4096 w=12 BETWEEN sel={sel}", n, 12 * 8 * W + 8 * W + D * 8, tmin,
               tmed, cnt == exp, {"selectivity": round(cnt / n, 4)})
    dd.close()
    del codes, enc

    # ---- configs[3]: dictionary int32, D in {256,4096,40000}, IN K in {4,16} + gather of selected ----
    for D in (256, 4096, 40000):
        bw = capi.dict_bit_width(D)
        dict_vals = np.sort(rng.choice(np.arange(-2 ** 30, 2 ** 30, 7), D, replace=False)).astype(np.int32)
        codes = codes_column(capi, ips.synth.SEED_DICT, n, D)
        enc = capi.fle_encode(codes, bw)
        dd = capi.Dict(dict_vals.view(np.uint8), capi.T_INT32)
        for K in (4, 16):
            present_idx = rng.choice(D, K // 2, replace=False)
            lits = np.concatenate([dict_vals[present_idx], dict_vals[present_idx] + 1]).astype(np.int32)
            res = {}
            def f():
                res["r"] = dd.scan(enc, n, bw, capi.OP_IN, lits)
            tmin, tmed = timeit(f, reps=6)
            bitmap, bvals, counts = res["r"]
            nsel = int(counts.to(torch.int64).sum().item())
            exp = int(torch.isin(codes, torch.tensor(np.sort(present_idx), device=dev, dtype=torch.int32)).sum().item())
            report(f"configs[3] dict int32 D={D} w={bw} IN K={K} fused scan+gather", n,
                   bw * 8 * W + 8 * W + 4 * nsel + D * 4, tmin, tmed, nsel == exp,
                   {"selectivity": round(nsel / n, 5)})
        dd.close()
        del codes, enc

    # ---- configs[4]: TPC-H Q6 shape: shipdate D=2526 w=12, discount D=11 w=4, quantity D=50 w=6 ----
    nq = args.q6_rows
    Wq = (nq + 63) // 64
    cols_codes = []
    encs = []
    for seed, D, bw in ((ips.synth.SEED_Q6[0], 2526, 12), (ips.synth.SEED_Q6[1], 11, 4), (ips.synth.SEED_Q6[2], 50, 6)):
        cdz = codes_column(capi, seed, nq, D)
        encs.append(capi.fle_encode(cdz, bw))
        cols_codes.append(cdz)
    # shipdate in one of seven years: codes [365, 730); discount BETWEEN 0.05 AND 0.07: codes 5..7;
    # quantity < 24: codes < 23
    cols = [capi.fle_column(encs[0], 12), capi.fle_column(encs[1], 4), capi.fle_column(encs[2], 6)]
    L, AND = capi.leaf, capi.and_node
    nodes = [L(0, capi.OP_GE, 365), L(0, capi.OP_LT, 730), AND(), L(1, capi.OP_GE, 5), L(1, capi.OP_LT, 8), AND(),
             AND(), L(2, capi.OP_LT, 23), AND()]
    bm = torch.empty(Wq, dtype=torch.int64, device=dev)
    f = lambda: capi.eval_program(nodes, cols, nq, bitmap=bm)
    tmin, tmed = timeit(f)
    cnt = capi.bitmap_count(bm, nq)
    exp = int(((cols_codes[0] >= 365) & (cols_codes[0] < 730) & (cols_codes[1] >= 5) & (cols_codes[1] < 8)
               & (cols_codes[2] < 23)).sum().item())
    byts = (12 + 4 + 6) * 8 * Wq + 8 * Wq
    report("configs[4] TPC-H Q6 shape, 3 dictionary columns (w=12,4,6), fused program, 1 GPU", nq, byts, tmin,
           tmed, cnt == exp, {"selectivity": round(cnt / nq, 5)})
    # the same conjunction leaf by leaf (5 predicate launches + 4 bitmap ANDs), for comparison
    tmp = [torch.empty(Wq, dtype=torch.int64, device=dev) for _ in range(2)]
    def leafwise():
        capi.fle_pred(encs[0], nq, 12, capi.OP_GE, 365, bitmap=tmp[0])
        for (e, bw, op, c) in ((encs[0], 12, capi.OP_LT, 730), (encs[1], 4, capi.OP_GE, 5),
                               (encs[1], 4, capi.OP_LT, 8), (encs[2], 6, capi.OP_LT, 23)):
            capi.fle_pred(e, nq, bw, op, c, bitmap=tmp[1])
            capi.bitmap_and(tmp[0], tmp[1], nq)
    tmin, tmed = timeit(leafwise)
    report("configs[4] same conjunction, leaf-by-leaf launches + bitmap ANDs", nq, byts, tmin, tmed,
           torch.equal(tmp[0], bm))
    # the same scan carried through to row-major tuples: the selected rows' discount and quantity
    # (dictionary int32 values) are gathered against the bitmap and assembled into 8-byte tuples
    import ctypes as C
    dpage = (np.arange(11, dtype=np.int32) + 100)
    qpage = (np.arange(50, dtype=np.int32) + 1)
    dd_disc = capi.Dict(dpage.view(np.uint8), 2)
    dd_qty = capi.Dict(qpage.view(np.uint8), 2)
    lib = capi.lib()
    NQ = C.c_int64(nq)
    Pp = lambda t: C.c_void_p(t.data_ptr())
    St = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    nbq = (nq + 2047) // 2048
    bv_d = torch.empty(nbq * 2048, dtype=torch.int32, device=dev)
    bv_q = torch.empty(nbq * 2048, dtype=torch.int32, device=dev)
    cnts = torch.empty(nbq, dtype=torch.int32, device=dev)
    lib.ips_assemble_workspace_bytes.restype = C.c_size_t
    wsz = int(lib.ips_assemble_workspace_bytes(NQ, 0)) + 64
    ws = torch.empty(wsz, dtype=torch.uint8, device=dev)
    tot = torch.zeros(1, dtype=torch.int64, device=dev)
    tcols = (capi.TupleColumn * 2)()
    tcols[0].d_batch_values, tcols[0].value_width, tcols[0].tuple_offset = bv_d.data_ptr(), 4, 0
    tcols[1].d_batch_values, tcols[1].value_width, tcols[1].tuple_offset = bv_q.data_ptr(), 4, 4
    tuples = torch.empty(max(cnt, 1) * 8 + 64, dtype=torch.uint8, device=dev)

    def pipeline():
        capi.eval_program(nodes, cols, nq, bitmap=bm)
        lib.ips_dict_select(dd_disc.h, Pp(encs[1]), NQ, 4, Pp(bm), Pp(bv_d), Pp(cnts), St)
        lib.ips_dict_select(dd_qty.h, Pp(encs[2]), NQ, 6, Pp(bm), Pp(bv_q), Pp(cnts), St)
        lib.ips_assemble_tuples(tcols, 2, Pp(cnts), NQ, 8, None, Pp(tuples), Pp(tot), Pp(ws), St)
    tmin, tmed = timeit(pipeline)
    ntup = int(tot.item())
    tview = tuples[:ntup * 8].view(torch.int32).view(ntup, 2)
    selmask = ((cols_codes[0] >= 365) & (cols_codes[0] < 730) & (cols_codes[1] >= 5) & (cols_codes[1] < 8)
               & (cols_codes[2] < 23))
    ok = (ntup == cnt and torch.equal(tview[:, 0], (cols_codes[1][selmask] + 100).to(torch.int32))
          and torch.equal(tview[:, 1], (cols_codes[2][selmask] + 1).to(torch.int32)))
    byts_pipe = byts + (4 + 6) * 8 * Wq + 2 * 8 * Wq + ntup * (8 + 8 + 8)
    report("configs[4] Q6 scan + late materialisation of 2 columns into 8-byte tuples (4 calls)", nq,
           byts_pipe, tmin, tmed, bool(ok), {"tuples": ntup})
    dd_disc.close(); dd_qty.close()
    del bv_d, bv_q, tuples

    # an OR of two conjunctions over the same columns: two bitmaps live at once
    OR = capi.or_node
    nodes2 = [L(0, capi.OP_GE, 365), L(0, capi.OP_LT, 730), AND(), L(1, capi.OP_LT, 3), AND(),
              L(2, capi.OP_GE, 40), L(1, capi.OP_GE, 9), AND(), OR()]
    exp2 = int((((cols_codes[0] >= 365) & (cols_codes[0] < 730) & (cols_codes[1] < 3))
                | ((cols_codes[2] >= 40) & (cols_codes[1] >= 9))).sum().item())
    byts2 = (12 + 4 + 6 + 4) * 8 * Wq + 8 * Wq
    for label, strat in (("per-operand launches + one bitmap OR", capi.PROGRAM_AUTO),
                         ("one-launch program kernel", capi.PROGRAM_ONE_LAUNCH)):
        capi.set_program_strategy(strat)
        f2 = lambda: capi.eval_program(nodes2, cols, nq, bitmap=bm)
        tmin, tmed = timeit(f2)
        report(f"configs[4] (A and B and C) or (D and E), {label}", nq, byts2, tmin, tmed,
               capi.bitmap_count(bm, nq) == exp2)
    capi.set_program_strategy(capi.PROGRAM_AUTO)
    json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out",
                                     "configs_bench.json"), "w"), indent=1)


if __name__ == "__main__":
    main()

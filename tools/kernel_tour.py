#!/usr/bin/env python3
"""Profiling target: every kernel family of libips_hip.so a few times at 2^28 rows (Q6: 600 M),
each with its algorithmic bytes, so that `rocprofv3 --kernel-trace --stats -- python3
tools/kernel_tour.py` yields per-kernel durations for pred / decode / encode / PLAIN / program /
expand / compress / batches / tuples / dictionary / nullable next to the headline scan.  Prints one
JSON line per entry: {"op", "kernel" (substring of the kernel name), "bytes", "us_event"}."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

ips = entry.load_package()
capi = ips.capi
lib = capi.lib()
dev = torch.device("cuda")
n = int(os.environ.get("IPS_TOUR_ROWS", str(1 << 28)))
W = (n + 63) // 64
REPS = 5


MULTS = [int(x) for x in os.environ.get("IPS_TOUR_MULTS", "").split(",") if x]


def run(op, kernel, byts, fn):
    if MULTS:  # dev builds (-DIPS_DEV_KNOBS read IPS_GRID_MULT on every call): every multiplier, interleaved
        res = {m: [] for m in MULTS}
        for _ in range(3):
            for m in MULTS:
                os.environ["IPS_GRID_MULT"] = str(m)
                fn()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(REPS):
                    fn()
                b.record()
                torch.cuda.synchronize()
                res[m].append(a.elapsed_time(b) * 1e3 / REPS)
        os.environ.pop("IPS_GRID_MULT", None)
        print(f"{op[:86]:86s} " + "  ".join(f"x{m}: {sorted(res[m])[1]:7.1f}" for m in MULTS), flush=True)
        return
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(REPS):
        fn()
    b.record()
    torch.cuda.synchronize()
    print(json.dumps({"op": op, "kernel": kernel, "bytes": int(byts), "us_event": round(a.elapsed_time(b) * 1e3 / REPS, 1)}),
          flush=True)


for bw in (32, 16, 8):
    vals = capi.synth_u32(ips.synth.SEED_HEADLINE, n, bw)
    enc = capi.fle_encode(vals, bw)
    c = ips.synth.lt_constant(bw)
    outs = capi.alloc_scan_outputs(n, dev)
    ow = 1 if bw <= 8 else 2 if bw <= 16 else 4
    dec = torch.empty(n, dtype={1: torch.uint8, 2: torch.int16, 4: torch.int32}[ow], device=dev)
    run(f"fle_pred w={bw} LT", f"fle_pred32_early_kernel<32, false>" if bw == 32 else f"fle_pred_w_kernel<{bw}, 0>", W * 8 * (bw + 1),
        lambda: capi.fle_pred(enc, n, bw, capi.OP_LT, c, bitmap=outs[0]))
    run(f"fle_pred w={bw} BETWEEN", f"fle_pred32_early_kernel<32, true>" if bw == 32 else f"fle_pred_w_kernel<{bw}, 1>", W * 8 * (bw + 1),
        lambda: capi.eval_program([capi.leaf(0, capi.OP_GE, c // 2), capi.leaf(0, capi.OP_LT, c), capi.and_node()],
                                  [capi.fle_column(enc, bw)], n, bitmap=outs[0]))
    capi.fle_scan(enc, n, bw, capi.OP_LT, c, outputs=outs)
    nsel = int(outs[2].to(torch.int64).sum().item())
    run(f"fle_scan w={bw} LT @10%", f"fle_scan_kernel<{bw}, 0, 0>", W * 8 * (bw + 1) + 4 * nsel,
        lambda: capi.fle_scan(enc, n, bw, capi.OP_LT, c, outputs=outs))
    bm = outs[0].clone()
    run(f"fle_select w={bw} @10%", f"fle_scan_kernel<{bw}, 1, 0>", W * 8 * (bw + 1) + 4 * nsel,
        lambda: capi.fle_select(enc, n, bw, bm, outputs=outs))
    run(f"fle_decode w={bw}", f"fle_decode_kernel<{bw},", W * 8 * bw + n * ow,
        lambda: capi.fle_decode(enc, n, bw, ow, out=dec))
    run(f"fle_encode w={bw}", f"fle_encode_kernel<{bw},", W * 8 * bw + n * 4, lambda: capi.fle_encode(vals, bw, out=enc))
    del vals, enc, outs, dec, bm

# PLAIN int64
x = capi.synth_u32(0x5EED0003, n, 32).to(torch.int64) & 0xFFFFFFFF
page = ((capi.synth_u32(0x5EED1003, n, 8).to(torch.int64) << 32) | x)
del x
lo, hi = int(0.45 * (1 << 40)), int(0.55 * (1 << 40))
bm = torch.empty(W, dtype=torch.int64, device=dev)
nodes = [capi.plain_leaf(0, capi.OP_GE, np.int64(lo), capi.T_INT64), capi.plain_leaf(0, capi.OP_LE, np.int64(hi), capi.T_INT64),
         capi.and_node()]
cols = [capi.plain_column(page, capi.T_INT64)]
run("plain_pred int64 BETWEEN @10%", "plain_tile_kernel<long, unsigned long, false>", 8 * n + n // 8, lambda: capi.eval_program(nodes, cols, n, bitmap=bm))
res = {}


def pscan():
    res["r"] = capi.plain_scan(page, n, capi.T_INT64, capi.OP_GE, np.int64(lo), op2=capi.OP_LE, literal2=np.int64(hi))


pscan()
nsel = int(res["r"][2].to(torch.int64).sum().item())
run("plain_scan int64 BETWEEN @10%", "plain_tile_kernel<long, unsigned long, true>", 8 * n + n // 8 + 8 * nsel, pscan)
del res
run("plain_select int64 @10%", "plain_select_kernel<unsigned long", n // 8 + 16 * nsel, lambda: capi.plain_select(page, n, capi.T_INT64, bm))
del page

# bitmaps, expand / compress, batches, tuples
vals = capi.synth_u32(ips.synth.SEED_HEADLINE, n, 32)
enc = capi.fle_encode(vals, 32)
outs = capi.alloc_scan_outputs(n, dev)
capi.fle_scan(enc, n, 32, capi.OP_LT, ips.synth.lt_constant(32), outputs=outs)
bm10, bvals, counts = outs
nsel = int(counts.to(torch.int64).sum().item())
bm50 = capi.fle_pred(enc, n, 32, capi.OP_GE, 1 << 31).clone()
acc = bm10.clone()
P = lambda t: C.c_void_p(t.data_ptr())
N = C.c_int64(n)
S = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = torch.empty(max(int(lib.ips_expand_workspace_bytes(N)), int(lib.ips_batches_workspace_bytes(N)),
                     int(lib.ips_assemble_workspace_bytes(N, 2)), 16) + 256, dtype=torch.uint8, device=dev)
out_bm = torch.empty(W + 2, dtype=torch.int64, device=dev)
cnt = torch.zeros(3, dtype=torch.int64, device=dev)
dense = torch.empty(n, dtype=torch.int32, device=dev)
run("bitmap_and", "bitmap_binop_kernel", 3 * W * 8, lambda: lib.ips_bitmap_and(P(acc), P(bm50), N, S))
run("bitmap_count", "bitmap_count_kernel", W * 8, lambda: lib.ips_bitmap_count(P(bm10), N, P(cnt), S))
run("bitmap_expand (root 50%)", "expand_kernel<0, 0>", 3 * W * 8, lambda: lib.ips_bitmap_expand(P(bm50), P(bm10), N, P(out_bm), P(ws), S))
run("bitmap_compress (mask 50%)", "compress_kernel<0, 0>", 3 * W * 8,
    lambda: lib.ips_bitmap_compress(P(bm50), P(bm10), N, P(out_bm), P(cnt), P(ws), S))
run("batches_compact @10%", "batches_compact_kernel", 8 * nsel + counts.numel() * 4,
    lambda: lib.ips_batches_compact(P(bvals), P(counts), N, 4, P(dense), P(cnt), P(ws), S))
tc = (capi.TupleColumn * 3)()
for i in range(3):
    tc[i].d_batch_values, tc[i].value_width, tc[i].tuple_offset = bvals.data_ptr(), 4, 4 * i
tuples = torch.empty(nsel * 16 + 64, dtype=torch.uint8, device=dev)
run("assemble_tuples 3 x int32 -> 16 B", "assemble_small_kernel<16>", nsel * 28 + counts.numel() * 4,
    lambda: lib.ips_assemble_tuples(tc, 3, P(counts), N, 16, None, P(tuples), P(cnt), P(ws), S))
del vals, enc, outs, bm50, acc, tuples, dense

# dictionary decode with a dictionary shared by one workgroup per CU (beyond 32 KiB), and the largest one
for Db, bwb, kname in ((16384, 14, "fle_decode_kernel<14, 4, 4, 16"), (40000, 16, "fle_decode_kernel<16, 4, 4, 8, true, true>")):
    rngb = np.random.default_rng(Db)
    dvb = np.sort(rngb.choice(np.arange(-2 ** 30, 2 ** 30, 7), Db, replace=False)).astype(np.int32)
    cb = ((capi.synth_u32(ips.synth.SEED_DICT, n, 32).to(torch.int64) & 0xFFFFFFFF) % Db).to(torch.int32)
    eb = capi.fle_encode(cb, bwb)
    ddb = capi.Dict(dvb.view(np.uint8), capi.T_INT32)
    run(f"dict_decode D={Db} w={bwb} int32 (shared LDS dictionary)", kname, bwb * 8 * W + 4 * n, lambda: ddb.decode(eb, n, bwb))
    ddb.close()
    del cb, eb

# dictionary: decode (gather), IN scan, encode
D = 4096
rng = np.random.default_rng(4)
dict_vals = np.sort(rng.choice(np.arange(-2 ** 30, 2 ** 30, 7), D, replace=False)).astype(np.int32)
codes = ((capi.synth_u32(ips.synth.SEED_DICT, n, 32).to(torch.int64) & 0xFFFFFFFF) % D).to(torch.int32)
enc = capi.fle_encode(codes, 12)
dd = capi.Dict(dict_vals.view(np.uint8), capi.T_INT32)
run("dict_decode D=4096 w=12 int32", "fle_decode_kernel<12, 4, 4, 4", 12 * 8 * W + 4 * n, lambda: dd.decode(enc, n, 12))
present = rng.choice(D, 8, replace=False)
lits = np.concatenate([dict_vals[present], dict_vals[present] + 1]).astype(np.int32)
r0 = dd.scan(enc, n, 12, capi.OP_IN, lits)
nsel = int(r0[2].to(torch.int64).sum().item())
del r0
run("dict_scan D=4096 w=12 IN K=16", "fle_scan_kernel<12, 2, 4>", 12 * 8 * W + 8 * W + 4 * nsel, lambda: dd.scan(enc, n, 12, capi.OP_IN, lits))
plain = torch.from_numpy(dict_vals).cuda()[codes.to(torch.int64)]
run("dict_encode D=4096 int32 (insert + lookup + FLE)", "dict_", 2 * 4 * n + 12 * 8 * W, lambda: capi.dict_encode(plain, capi.T_INT32))
dd.close()
del codes, enc, plain

# nullable leaf
nn = capi.synth_u32(0x5EED0D1, n, 32)
is_set = (nn.to(torch.int64) & 0xFFFFFFFF) >= int(0.1 * (1 << 32))
del nn
defs = capi.fle_encode(is_set.to(torch.int32), 1)
k = int(is_set.sum().item())
del is_set
venc = capi.fle_encode(capi.synth_u32(0x5EED0D2, k, 12), 12)
n_data = ((k + 63) // 64) * 64
wsn = capi.nullable_workspace(n, dev)
run("nullable leaf w=12, 10% NULL (counts + leaf)", "fle_leaf_kernel<12, 0>", W * 16 + n_data // 64 * 96,
    lambda: capi.fle_pred_nullable(defs, 1, 1, n, venc, n_data, 12, capi.OP_LT, 409, bitmap=bm, workspace=wsn))
# late materialisation of the rows the leaf selected (raw values): counts + select kernel
lib.ips_select_nullable_workspace_bytes.restype = C.c_size_t
ws_sn = torch.empty(int(lib.ips_select_nullable_workspace_bytes(N, C.c_int64(n_data), 4)) + 16, dtype=torch.uint8, device=dev)
sel_bm = bm.clone()
n_sel_nn = capi.bitmap_count(sel_bm, n)
dense_sn = torch.empty(n_sel_nn + 64, dtype=torch.int32, device=dev)
flags_sn = torch.empty(W, dtype=torch.int64, device=dev)
cnt_sn = torch.zeros(3, dtype=torch.int64, device=dev)
run("select_nullable w=12, 10% NULL, the leaf's selection (counts + select)", "fle_select_nullable_kernel<12, 0>",
    W * 16 + n_data // 64 * 96 + 4 * n_sel_nn + n_sel_nn // 8,
    lambda: capi._ck(lib.ips_dict_select_nullable(None, P(defs), 1, 1, N, P(venc), C.c_int64(n_data), 12, P(sel_bm), P(dense_sn),
                                                  P(flags_sn), P(cnt_sn), P(ws_sn), S)))
del defs, venc, wsn, ws_sn, dense_sn, flags_sn, sel_bm

# Q6 program
q6 = ips.q6
nq = q6.ROWS if n >= (1 << 28) else n
codes = [q6.codes_gpu(capi, c, nq) for c in range(3)]
encs = [capi.fle_encode(codes[c], q6.COLUMNS[c][3]) for c in range(3)]
del codes
nodes, cols = q6.program(capi, encs)
bmq = torch.empty((nq + 63) // 64, dtype=torch.int64, device=dev)
run("Q6 conjunction, 3 columns, one-pass chain (1 launch)", "fle_chain_w_kernel<6, 16>", q6.algorithmic_bytes(nq),
    lambda: capi.eval_program(nodes, cols, nq, bitmap=bmq))


def q6_per_operand():
    capi.set_program_strategy(capi.PROGRAM_PER_OPERAND)
    capi.eval_program(nodes, cols, nq, bitmap=bmq)
    capi.set_program_strategy(capi.PROGRAM_AUTO)


run("Q6 conjunction, 3 columns, per-operand plan (3 launches)", "fle_pred_w_kernel<12, 1>", q6.algorithmic_bytes(nq), q6_per_operand)

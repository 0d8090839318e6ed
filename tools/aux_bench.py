#!/usr/bin/env python3
"""dev tool: the auxiliary kernels (bitmap algebra, expand/compress, batch compaction, tuple
assembly) against the bytes they have to move.  2^28 rows, bitmap at 10 % selectivity."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402
from tools.kbench import timeit  # noqa: E402

ips = entry.load_package()
capi = ips.capi
lib = capi.lib()
n = 1 << 28
W = n // 64
dev = torch.device("cuda")
P = lambda t: C.c_void_p(t.data_ptr())
N = C.c_int64(n)
S = C.c_void_p(torch.cuda.current_stream().cuda_stream)

vals = capi.synth_u32(ips.synth.SEED_HEADLINE, n, 32)
enc = capi.fle_encode(vals, 32)
c10 = int(0.1 * (1 << 32))
outs = capi.alloc_scan_outputs(n, dev)
capi.fle_scan(enc, n, 32, capi.OP_LT, c10, outputs=outs)
bm, bvals, counts = outs
nsel = int(counts.to(torch.int64).sum().item())
bm2 = capi.fle_pred(enc, n, 32, capi.OP_GE, 1 << 31).clone()          # 50 % mask
ws = torch.empty(max(int(lib.ips_expand_workspace_bytes(N)), int(lib.ips_batches_workspace_bytes(N)),
                     int(lib.ips_assemble_workspace_bytes(N, 2)), 16), dtype=torch.uint8, device=dev)
out_bm = torch.empty(W + 2, dtype=torch.int64, device=dev)
cnt = torch.zeros(3, dtype=torch.int64, device=dev)
dense = torch.empty(n, dtype=torch.int32, device=dev)
acc = bm.clone()


def show(name, byts, f):
    tmin, tmed = timeit(f)
    print(f"{name:44s} {byts / 1e6:9.1f} MB  min {tmin * 1e3:7.1f} us  med {tmed * 1e3:7.1f} us  "
          f"{byts / tmed / 1e6:7.0f} GB/s", flush=True)


show("bitmap_and (2 reads + 1 write)", 3 * W * 8, lambda: lib.ips_bitmap_and(P(acc), P(bm2), N, S))
show("bitmap_count (1 read)", W * 8, lambda: lib.ips_bitmap_count(P(bm), N, P(cnt), S))
show("bitmap_fill (1 write)", W * 8, lambda: lib.ips_bitmap_fill(P(out_bm), N, 1, S))
show("bitmap_expand (root 50 %, sub) ", 3 * W * 8, lambda: lib.ips_bitmap_expand(P(bm2), P(bm), N, P(out_bm), P(ws), S))
show("bitmap_compress (mask 50 %, src)", 3 * W * 8, lambda: lib.ips_bitmap_compress(P(bm2), P(bm), N, P(out_bm), P(cnt), P(ws), S))
show("batches_compact (10 % of 2^28 values)", 8 * nsel + counts.numel() * 4,
     lambda: lib.ips_batches_compact(P(bvals), P(counts), N, 4, P(dense), P(cnt), P(ws), S))
# three REQUIRED int32 columns selected by the same bitmap -> 16-byte tuples
cols = (capi.TupleColumn * 3)()
for i in range(3):
    cols[i].d_batch_values = bvals.data_ptr()
    cols[i].value_width = 4
    cols[i].tuple_offset = 4 * i
tuples = torch.empty(nsel * 16 + 64, dtype=torch.uint8, device=dev)
show("assemble_tuples (3 int32 cols -> 16 B tuples)", nsel * (12 + 16) + counts.numel() * 4,
     lambda: lib.ips_assemble_tuples(cols, 3, P(counts), N, 16, None, P(tuples), P(cnt), P(ws), S))
show("fle_select given bitmap (w=32, 10 %)", W * 8 * 33 + 4 * nsel,
     lambda: lib.ips_fle_select(P(enc), N, 32, P(bm), P(bvals), P(counts), S))
page64 = torch.arange(n, dtype=torch.int64, device=dev)
bv64 = torch.empty(n, dtype=torch.int64, device=dev)
show("plain_select int64 given bitmap (10 %)", W * 8 + nsel * 16,
     lambda: lib.ips_plain_select(P(page64), N, 3, P(bm), P(bv64), P(counts), S))

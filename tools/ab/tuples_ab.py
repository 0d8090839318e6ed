#!/usr/bin/env python3
"""Dev tool: ips_assemble_tuples for wider tuples (LDS image path): 2^28 rows selected at 10 %,
(int32, int64, int32) -> 24-byte tuples, 4 x int32 -> 32-byte tuples, 8 x int32 -> 64-byte tuples."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as entry  # noqa: E402
from tools.kbench import timeit  # noqa: E402

ips = entry.load_package()
capi = ips.capi
lib = capi.lib()
n = 1 << 28
dev = torch.device("cuda")
P = lambda t: C.c_void_p(t.data_ptr())
N = C.c_int64(n)
S = C.c_void_p(torch.cuda.current_stream().cuda_stream)
vals = capi.synth_u32(ips.synth.SEED_HEADLINE, n, 32)
enc = capi.fle_encode(vals, 32)
outs = capi.alloc_scan_outputs(n, dev)
capi.fle_scan(enc, n, 32, capi.OP_LT, int(0.1 * (1 << 32)), outputs=outs)
bm, bvals, counts = outs
nsel = int(counts.to(torch.int64).sum().item())
bv64 = torch.arange(n, dtype=torch.int64, device=dev)
ws = torch.empty(int(lib.ips_assemble_workspace_bytes(N, 2)) + 256, dtype=torch.uint8, device=dev)
cnt = torch.zeros(3, dtype=torch.int64, device=dev)
for name, layout, ts in (("int32,int64,int32 -> 24 B", ((4, 0), (8, 8), (4, 16)), 24),
                         ("4 x int32 -> 32 B", ((4, 0), (4, 8), (4, 16), (4, 24)), 32),
                         ("8 x int32 -> 64 B", tuple((4, 8 * i) for i in range(8)), 64)):
    cols = (capi.TupleColumn * len(layout))()
    byts = nsel * ts + counts.numel() * 4
    for i, (w, off) in enumerate(layout):
        cols[i].d_batch_values = (bvals if w == 4 else bv64).data_ptr()
        cols[i].value_width = w
        cols[i].tuple_offset = off
        byts += nsel * w
    tuples = torch.empty(nsel * ts + 64, dtype=torch.uint8, device=dev)
    tmin, tmed = timeit(lambda: lib.ips_assemble_tuples(cols, len(layout), P(counts), N, ts, None, P(tuples), P(cnt), P(ws), S))
    print(f"assemble {name:28s} {byts / 1e6:8.1f} MB  min {tmin * 1e3:7.1f} us  med {tmed * 1e3:7.1f} us  {byts / tmed / 1e6:6.0f} GB/s", flush=True)

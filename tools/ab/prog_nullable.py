#!/usr/bin/env python3
"""dev tool: a conjunction of a REQUIRED and two OPTIONAL dictionary columns (2^28 rows) through
ips_eval_program: launches per call and median us."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as entry  # noqa: E402
from tools.kbench import timeit  # noqa: E402

capi = entry.load_package().capi
n = 1 << 28
dev = torch.device("cuda")
a = capi.fle_encode(capi.synth_u32(0x5EED0A1, n, 12), 12)
cols = [capi.fle_column(a, 12)]
keep = [a]
for seed, bw in ((0x5EED0B1, 6), (0x5EED0C1, 9)):
    nn = capi.synth_u32(seed, n, 32)
    is_set = (nn.to(torch.int64) & 0xFFFFFFFF) >= int(0.1 * (1 << 32))
    defs = capi.fle_encode(is_set.to(torch.int32), 1)
    k = int(is_set.sum().item())
    del nn, is_set
    enc = capi.fle_encode(capi.synth_u32(seed + 1, k, bw), bw)
    cols.append(capi.nullable_fle_column(defs, 1, 1, enc, bw, k))
    keep += [defs, enc]
L, AND = capi.leaf, capi.and_node
nodes = [L(0, capi.OP_LT, 2000), L(1, capi.OP_GE, 10), AND(), L(2, capi.OP_LT, 300), AND()]
bm = torch.empty(n // 64, dtype=torch.int64, device=dev)
ws = torch.empty(capi.program_workspace_bytes(nodes, cols, n) + 64, dtype=torch.uint8, device=dev)
tmin, tmed = timeit(lambda: capi.eval_program(nodes, cols, n, bitmap=bm, workspace=ws), reps=20)
print(f"A(w12) and B?(w6) and C?(w9): med {tmed*1e3:.1f} us min {tmin*1e3:.1f} us  count {capi.bitmap_count(bm, n)}", flush=True)

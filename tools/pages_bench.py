#!/usr/bin/env python3
"""dev tool: 256 separate 2^20-row pages through ips_fle_scan_pages vs one contiguous 2^28-row
ips_fle_scan vs 256 single-page launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402
from tools.kbench import timeit  # noqa: E402

ips = entry.load_package()
capi = ips.capi
dev = torch.device("cuda")
bw, n_pages, rows = 32, 256, 1 << 20
n = n_pages * rows
c = int(0.1 * (1 << 32))
vals = capi.synth_u32(ips.synth.SEED_HEADLINE, n, bw)
enc = capi.fle_encode(vals, bw)
outs = capi.alloc_scan_outputs(n, dev)
tmin, tmed = timeit(lambda: capi.fle_scan(enc, n, bw, capi.OP_LT, c, outputs=outs))
print(f"one contiguous 2^28-row scan:        min {tmin*1e3:7.1f} us  med {tmed*1e3:7.1f} us")
wpp = rows // 64 * bw
pages = []
for p in range(n_pages):
    e = enc[p * wpp:(p + 1) * wpp].clone()          # separate allocations
    pages.append((e, rows, capi.alloc_scan_outputs(rows, dev)))
plist = capi.make_page_list(pages)
tmin, tmed = timeit(lambda: capi.fle_scan_pages(plist, bw, capi.OP_LT, c))
print(f"256 pages, ips_fle_scan_pages:        min {tmin*1e3:7.1f} us  med {tmed*1e3:7.1f} us")
def one_by_one():
    for e, r, o in pages:
        capi.fle_scan(e, r, bw, capi.OP_LT, c, outputs=o)
tmin, tmed = timeit(one_by_one, reps=5, warm=1)
print(f"256 pages, one ips_fle_scan each:     min {tmin*1e3:7.1f} us  med {tmed*1e3:7.1f} us")
# same result?
capi.fle_scan(enc, n, bw, capi.OP_LT, c, outputs=outs)
capi.fle_scan_pages(plist, bw, capi.OP_LT, c)
ok = all(torch.equal(pages[p][2][0][:rows // 64], outs[0][p * rows // 64:(p + 1) * rows // 64]) for p in range(n_pages))
print("bitmaps equal:", ok)

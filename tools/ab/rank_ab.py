#!/usr/bin/env python3
"""Dev tool: ips_bitmap_expand / ips_bitmap_compress at 2^28 rows, 10 % zeros in the root."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as entry  # noqa: E402
from tools.nullable_bench import timeit  # noqa: E402


def main():
    capi = entry.load_package().capi
    n = 1 << 28
    nn = capi.synth_u32(0x5EED0D1, n, 32)
    root_bits = (nn.to(torch.int64) & 0xFFFFFFFF) >= int(0.1 * (1 << 32))
    w = root_bits.view(-1, 64).to(torch.int64)
    root = (w << torch.arange(64, device=w.device, dtype=torch.int64)).sum(dim=1)
    del w, nn, root_bits
    sub = capi.synth_u32(0x5EED0D3, n // 32, 32).view(torch.int64).clone()
    for name, fn in (("expand", lambda: capi.bitmap_expand(root, sub, n)),
                     ("compress", lambda: capi.bitmap_compress(root, sub, n))):
        tmin, tmed = timeit(fn)
        print(f"{name:9s} rows={n} min {tmin*1e6:7.1f} us med {tmed*1e6:7.1f} us", flush=True)


if __name__ == "__main__":
    main()

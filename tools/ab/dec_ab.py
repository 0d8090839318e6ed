#!/usr/bin/env python3
"""Dev tool: decode of narrow columns to dwords (ips_fle_decode out_width 4) and dictionary decode."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as entry  # noqa: E402
from tools.nullable_bench import timeit  # noqa: E402


def main():
    ips = entry.load_package()
    capi = ips.capi
    n = 1 << 28
    for bw in (8, 12, 16):
        vals = capi.synth_u32(ips.synth.SEED_HEADLINE, n, bw)
        enc = capi.fle_encode(vals, bw)
        out = torch.empty(n, dtype=torch.int32, device=vals.device)
        tmin, tmed = timeit(lambda: capi.fle_decode(enc, n, bw, 4, out=out))
        assert torch.equal(out, vals)
        byts = n // 64 * bw * 8 + 4 * n
        print(f"decode w={bw:2d} -> dwords    min {tmin*1e6:7.1f} us med {tmed*1e6:7.1f} us {byts/tmed/1e9:7.1f} GB/s {byts/tmed/8e12:5.3f}", flush=True)
        del out
        D = min(1 << bw, 40000) if bw < 16 else 40000
        codes = (vals.to(torch.int64) & 0xFFFFFFFF) % D
        enc2 = capi.fle_encode(codes.to(torch.int32), bw)
        for t, npt, tt in ((capi.T_INT32, np.int32, torch.int32), (capi.T_INT64, np.int64, torch.int64)):
            dv = np.sort(np.random.default_rng(bw).choice(np.arange(-2 ** 30, 2 ** 30, 3), D, replace=False)).astype(npt)
            dd = capi.Dict(dv.view(np.uint8), t)
            tmin, tmed = timeit(lambda: dd.decode(enc2, n, bw), reps=10)
            got, bad = dd.decode(enc2, n, bw)
            assert int(bad.item()) == 0
            assert torch.equal(got, torch.from_numpy(dv).to(got.device)[codes])
            byts = n // 64 * bw * 8 + dv.itemsize * n
            print(f"dict decode w={bw:2d} D={D:5d} {npt.__name__} min {tmin*1e6:7.1f} us med {tmed*1e6:7.1f} us {byts/tmed/1e9:7.1f} GB/s {byts/tmed/8e12:5.3f}", flush=True)
            dd.close()
            del got
        del vals, enc, codes, enc2
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

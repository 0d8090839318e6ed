#!/usr/bin/env python3
"""Dev tool: predicate on an OPTIONAL column, 2^28 rows -- the fused nullable leaf
(ips_fle_pred_nullable) against the composed path (def == max_def, count, data predicate, expand).
Algorithmic bytes: definition levels (n/8) + data blocks of the non-NULL rows + one bitmap (n/8)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402


def timeit(fn, reps=20, warm=3):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1 << 28)
    ap.add_argument("--bw", default="12")
    ap.add_argument("--nulls", default="0.1")
    args = ap.parse_args()
    capi = entry.load_package().capi
    n = args.rows
    dev = torch.device("cuda")
    for null_frac in [float(x) for x in args.nulls.split(",")]:
        nn = capi.synth_u32(0x5EED0D1, n, 32)
        is_set = (nn.to(torch.int64) & 0xFFFFFFFF) >= int(null_frac * (1 << 32))
        del nn
        defs = capi.fle_encode(is_set.to(torch.int32), 1)
        k = int(is_set.sum().item())
        del is_set
        for bw in [int(x) for x in args.bw.split(",")]:
            vals = capi.synth_u32(0x5EED0D2, k, bw)
            enc = capi.fle_encode(vals, bw)
            del vals
            n_data = ((k + 63) // 64) * 64
            c = int(0.1 * (1 << bw))
            ws = capi.nullable_workspace(n, dev)
            bm = torch.empty((n + 63) // 64, dtype=torch.int64, device=dev)
            byts = (n + 63) // 64 * 8 * 2 + n_data // 64 * bw * 8
            f = lambda: capi.fle_pred_nullable(defs, 1, 1, n, enc, n_data, bw, capi.OP_LT, c, bitmap=bm, workspace=ws)
            tmin, tmed = timeit(f)
            print(f"nullable leaf  w={bw:2d} nulls={null_frac:4.2f} rows={n} min {tmin*1e6:7.1f} us med {tmed*1e6:7.1f} us "
                  f"{byts/tmed/1e9:7.1f} GB/s  frac {byts/tmed/8e12:5.3f}", flush=True)
            nonnull = torch.empty_like(bm)
            sub = torch.empty_like(bm)

            def composed():
                capi.fle_pred(defs, n, 1, capi.OP_EQ, 1, bitmap=nonnull)
                kk = capi.bitmap_count(nonnull, n)          # host sync, as the reference's count()
                capi.fle_pred(enc, kk, bw, capi.OP_LT, c, bitmap=sub)
                return capi.bitmap_expand(nonnull, sub, n)
            tmin2, tmed2 = timeit(composed, reps=10)
            assert torch.equal(composed(), bm)
            print(f"composed path  w={bw:2d} nulls={null_frac:4.2f} rows={n} min {tmin2*1e6:7.1f} us med {tmed2*1e6:7.1f} us "
                  f"{byts/tmed2/1e9:7.1f} GB/s", flush=True)
            del enc


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""dev tool: the nullable leaf (levels + data -> bitmap) on 2^28 rows, median us per call, for a
few widths / NULL fractions / predicate shapes.  A/B: IPS_NO_FUSED_LEAF=1 (predicate + expand as
separate launches) against the default (fle_leaf_kernel)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as entry  # noqa: E402


def med_us(fn, reps=15):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    capi = entry.load_package().capi
    n = 1 << 28
    dev = torch.device("cuda")
    for null_frac in (0.1, 0.5):
        nn = capi.synth_u32(0x5EED0D1, n, 32)
        is_set = (nn.to(torch.int64) & 0xFFFFFFFF) >= int(null_frac * (1 << 32))
        defs = capi.fle_encode(is_set.to(torch.int32), 1)
        k = int(is_set.sum().item())
        del nn, is_set
        ws = capi.nullable_workspace(n, dev)
        bm = torch.empty((n + 63) // 64, dtype=torch.int64, device=dev)
        for bw in (4, 12, 20, 32):
            enc = capi.fle_encode(capi.synth_u32(0x5EED0D2, k, bw), bw)
            n_data = ((k + 63) // 64) * 64
            c = int(0.1 * (1 << bw))
            alg = n / 8 * 2 + k * bw / 8
            t1 = med_us(lambda: capi.fle_pred_nullable(defs, 1, 1, n, enc, n_data, bw, capi.OP_LT, c, bitmap=bm, workspace=ws))
            t2 = med_us(lambda: capi.fle_pred_nullable(defs, 1, 1, n, enc, n_data, bw, capi.OP_IN, [1, 2, 3, c], bitmap=bm, workspace=ws))
            t3 = med_us(lambda: capi.fle_pred_nullable(defs, 1, 1, n, enc, n_data, bw, capi.OP_IN, [(c + 7 * i) % (1 << bw) for i in range(40)], bitmap=bm, workspace=ws)) if bw <= 16 else 0.0
            print(f"null {null_frac:.1f} w={bw:2d}: LT {t1:7.1f} us ({alg / t1 / 8e6 * 100:4.1f} %)   IN4 {t2:7.1f} us   IN40 {t3:7.1f} us", flush=True)
            del enc


if __name__ == "__main__":
    main()

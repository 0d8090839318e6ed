#!/usr/bin/env python3
"""Kernel micro-benchmarks (dev tool): time individual C-ABI entry points with HIP events.
usage: python tools/kbench.py [--rows N] [--bw 32,16,8] [--what pred,scan,decode,select,encode]"""
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
    return ts[0], ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1 << 28)
    ap.add_argument("--bw", default="32")
    ap.add_argument("--what", default="pred,scan,decode,select")
    ap.add_argument("--sel", default="0.1")
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    ips = entry.load_package()
    capi = ips.capi
    n = args.rows
    dev = torch.device("cuda")
    for bw in [int(x) for x in args.bw.split(",")]:
        vals = capi.synth_u32(ips.synth.SEED_HEADLINE, n, bw)
        enc = capi.fle_encode(vals, bw)
        blocks = (n + 63) // 64
        enc_bytes = blocks * bw * 8
        outs = capi.alloc_scan_outputs(n, dev)
        for sel in [float(s) for s in args.sel.split(",")]:
            c = min(int(sel * (1 << bw)), (1 << bw) - 1)
            op = capi.OP_LT if sel < 1.0 else capi.OP_LE
            for what in args.what.split(","):
                if what == "pred":
                    f = lambda: capi.fle_pred(enc, n, bw, op, c, bitmap=outs[0])
                    byts = enc_bytes + blocks * 8
                elif what == "scan":
                    f = lambda: capi.fle_scan(enc, n, bw, op, c, outputs=outs)
                    f()
                    nsel = int(outs[2].to(torch.int64).sum().item())
                    byts = enc_bytes + blocks * 8 + 4 * nsel
                elif what == "select":
                    capi.fle_pred(enc, n, bw, op, c, bitmap=outs[0])
                    nsel = capi.bitmap_count(outs[0], n)
                    f = lambda: capi.fle_select(enc, n, bw, outs[0], outputs=outs)
                    byts = enc_bytes + blocks * 8 + 4 * nsel
                elif what == "decode":
                    ow = 1 if bw <= 8 else 2 if bw <= 16 else 4
                    dt = {1: torch.uint8, 2: torch.int16, 4: torch.int32}[ow]
                    out = torch.empty(n, dtype=dt, device=dev)
                    f = lambda: capi.fle_decode(enc, n, bw, ow, out=out)
                    byts = enc_bytes + n * ow
                elif what == "decode4":
                    out = torch.empty(n, dtype=torch.int32, device=dev)
                    f = lambda: capi.fle_decode(enc, n, bw, 4, out=out)
                    byts = enc_bytes + n * 4
                elif what == "encode":
                    f = lambda: capi.fle_encode(vals, bw, out=enc)
                    byts = enc_bytes + n * 4
                else:
                    continue
                tmin, tmed = timeit(f, args.reps)
                print(f"w={bw:2d} sel={sel:4.2f} {what:8s} rows={n} min {tmin*1e3:8.1f} us  med {tmed*1e3:8.1f} us  "
                      f"{byts/tmin/1e6:8.1f} GB/s(min) {byts/tmed/1e6:8.1f} GB/s(med)  {n/tmed/1e6:8.1f} Grow/s", flush=True)
        del vals, enc, outs


if __name__ == "__main__":
    main()

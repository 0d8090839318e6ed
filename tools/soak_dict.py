#!/usr/bin/env python3
"""dev tool: randomized differential soak of the dictionary paths against numpy: decode (private,
shared and tail LDS copies from 2^20 rows on), fused scan + gather and select at every selectivity
up to all rows (dense store paths), 4- and 8-byte entries, dictionaries of 1 .. 40000 entries."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as entry  # noqa: E402

ips = entry.load_package()
capi = ips.capi
O = entry.load_oracle()
rng = np.random.default_rng(int(os.environ.get("IPS_SOAK_SEED", "20261007")))


def dev(a):
    a = np.ascontiguousarray(a)
    return torch.from_numpy(a.view(np.int64).copy()).cuda()


def dense_of(bv, c):
    c = c.cpu().numpy()
    bv = bv.cpu().numpy()
    return np.concatenate([bv[b * 2048: b * 2048 + c[b]] for b in range(len(c))]) if len(c) else bv[:0]


bad = 0
for it in range(int(os.environ.get("IPS_SOAK_ITERS", "120"))):
    t, npt = (capi.T_INT32, np.int32) if rng.random() < 0.5 else (capi.T_INT64, np.int64)
    D = int(rng.choice([1, 2, 17, 256, 1000, 4096, 5000, 9000, 16384, 30000, 40000]))
    bw = max(1, int(D - 1).bit_length())
    big = rng.random() < 0.25
    n = int((1 << 20) + rng.integers(0, 5000)) if big else int(rng.choice([1, 64, 2047, 2049, int(rng.integers(1, 100000))]))
    entries = np.sort(rng.choice(np.arange(-10 ** 9, 10 ** 9, 7), D, replace=False)).astype(npt)
    codes = rng.integers(0, D, n).astype(np.uint32)
    vals = entries[codes]
    dd = capi.Dict(entries.view(np.uint8), t)
    enc = dev(O.fle_encode(codes, bw))
    out, badidx = dd.decode(enc, n, bw)
    if int(badidx.item()) != 0 or not np.array_equal(out.cpu().numpy().astype(npt), vals):
        bad += 1
        print("decode mismatch", dict(D=D, bw=bw, n=n, t=npt.__name__), flush=True)
    frac = float(rng.choice([0.0, 0.01, 0.3, 0.7, 1.0]))
    if frac >= 1.0:
        op, lit, keep = capi.OP_LE, entries[-1], np.ones(n, bool)
    else:
        lit = entries[min(int(frac * D), D - 1)]
        op, keep = capi.OP_LT, vals < lit
    bitmap, bvals, counts = dd.scan(enc, n, bw, op, np.array([lit], dtype=npt))
    if not np.array_equal(dense_of(bvals, counts).astype(npt), vals[keep]):
        bad += 1
        print("scan mismatch", dict(D=D, bw=bw, n=n, frac=frac, t=npt.__name__), flush=True)
    sv, sc = dd.select(enc, n, bw, bitmap)
    if not np.array_equal(dense_of(sv, sc).astype(npt), vals[keep]):
        bad += 1
        print("select mismatch", dict(D=D, bw=bw, n=n, frac=frac, t=npt.__name__), flush=True)
    dd.close()
print("dict soak done, mismatches:", bad)
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""dev tool: randomized differential soak of pred / scan / select / decode against the oracle
(random widths, ragged sizes, skewed and uniform data, all six operators, IN lists on both paths)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as entry
ips = entry.load_package(); capi = ips.capi; O = entry.load_oracle()
rng = np.random.default_rng(int(os.environ.get("IPS_SOAK_SEED", "20261004")))
def words(t): return t.cpu().numpy().view(np.uint64)
bad = 0
for it in range(400):
    bw = int(rng.integers(1, 33))
    n = int(rng.choice([1, 63, 64, 65, 2047, 2048, 2049, 4097, int(rng.integers(1, 300000))]))
    mode = rng.random()
    if mode < 0.3:   # skewed: few distinct values
        pool = rng.integers(0, 1 << bw, int(rng.integers(1, 9)), dtype=np.uint64)
        vals = pool[rng.integers(0, len(pool), n)].astype(np.uint32)
    else:
        vals = rng.integers(0, 1 << bw, n, dtype=np.uint64).astype(np.uint32)
    ref_enc = O.fle_encode(vals, bw)
    enc = torch.from_numpy(np.ascontiguousarray(ref_enc).view(np.int64)).cuda()
    op = int(rng.integers(0, 6))
    if op == 5:
        k = int(rng.choice([1, 2, 5, 9, 10, 17, 40]))
        c = [int(x) for x in rng.choice(vals, k)] if rng.random() < 0.7 else [int(x) for x in rng.integers(0, 1 << bw, k)]
    else:
        c = int(vals[rng.integers(0, n)]) if rng.random() < 0.5 else int(rng.integers(0, 1 << bw))
    ref = O.fle_pred(ref_enc, n, bw, op, c)
    got = words(capi.fle_pred(enc, n, bw, op, c))
    if not np.array_equal(got, ref): bad += 1; print("pred mismatch", bw, n, op, c); continue
    bitmap, bvals, counts = capi.fle_scan(enc, n, bw, op, c)
    if not np.array_equal(words(bitmap), ref): bad += 1; print("scan bitmap mismatch", bw, n, op); continue
    cnt = counts.cpu().numpy(); bv = bvals.cpu().numpy().view(np.uint32)
    dense = np.concatenate([bv[b*2048:b*2048+cnt[b]] for b in range(len(cnt))]) if len(cnt) else np.zeros(0, np.uint32)
    exp = O.fle_select(ref_enc, n, bw, ref)
    if not np.array_equal(dense, exp): bad += 1; print("scan values mismatch", bw, n, op, len(dense), len(exp)); continue
    sv, sc = capi.fle_select(enc, n, bw, bitmap)
    scn = sc.cpu().numpy(); svn = sv.cpu().numpy().view(np.uint32)
    dense2 = np.concatenate([svn[b*2048:b*2048+scn[b]] for b in range(len(scn))]) if len(scn) else np.zeros(0, np.uint32)
    if not np.array_equal(dense2, exp): bad += 1; print("select mismatch", bw, n, op); continue
    ow = 1 if bw <= 8 else 2 if bw <= 16 else 4
    dec = capi.fle_decode(enc, n, bw).cpu().numpy().astype(np.uint32) if hasattr(capi, "fle_decode") else vals
    if not np.array_equal(dec[:n], vals): bad += 1; print("decode mismatch", bw, n)
print("soak done, mismatches:", bad)

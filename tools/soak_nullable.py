#!/usr/bin/env python3
"""dev tool: randomized differential soak of the OPTIONAL-column paths against the oracle's three
steps (levels == max_def, data predicate over the NOT-NULL count, IntersectBitset): the one-pass
leaf and the routes it hands back, level widths 1..3, clustered and uniform NULLs, data buffers
shorter than the NOT-NULL count, and ips_eval_program's and-into / or-into combine modes."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as entry  # noqa: E402

ips = entry.load_package()
capi = ips.capi
O = entry.load_oracle()
rng = np.random.default_rng(int(os.environ.get("IPS_SOAK_SEED", "20261006")))


def dev(a):
    a = np.ascontiguousarray(a)
    if a.size < 2:
        a = np.concatenate([a, np.zeros(2 - a.size, a.dtype)])
    return torch.from_numpy(a.view(np.int64).copy()).cuda()


def words(t):
    return t.cpu().numpy().view(np.uint64)


def bits_of(w, n):
    return np.unpackbits(np.ascontiguousarray(w).view(np.uint8), bitorder="little")[:n].astype(bool)


bad = 0
for it in range(int(os.environ.get("IPS_SOAK_ITERS", "250"))):
    bw = int(rng.integers(1, 33))
    n = int(rng.choice([1, 64, 65, 16384, 16385, 65536, 65537, int(rng.integers(1, 400000))]))
    max_def = int(rng.choice([1, 1, 1, 2, 3, 5]))
    def_bw = max(1, int(max_def).bit_length())
    style = rng.random()
    if style < 0.4:
        is_set = rng.random(n) >= rng.random()
    elif style < 0.8:   # runs of NULL / NOT NULL longer than a wave's 16384 rows
        is_set = np.zeros(n, bool)
        pos = 0
        while pos < n:
            ln = int(rng.integers(1, 40000))
            is_set[pos:pos + ln] = rng.random() < 0.5
            pos += ln
    else:
        is_set = np.full(n, rng.random() < 0.5)
    levels = np.where(is_set, max_def, rng.integers(0, max_def, n)).astype(np.uint32)
    k = int(is_set.sum())
    vals = rng.integers(0, 1 << bw, max(k, 1), dtype=np.uint64).astype(np.uint32)[:k]
    defs = O.fle_encode(levels, def_bw)
    enc = O.fle_encode(vals, bw) if k else np.zeros(2, np.uint64)
    d_defs, d_enc = dev(defs), dev(enc)
    n_data = k if rng.random() < 0.7 else int(rng.integers(0, k + 1))   # sometimes a short data buffer
    op = int(rng.integers(0, 6))
    if op == 5:
        kk = int(rng.choice([1, 3, 9, 16, 40]))
        c = [int(x) for x in (rng.choice(vals, kk) if k else rng.integers(0, 1 << bw, kk))]
    else:
        c = int(vals[rng.integers(0, k)]) if k and rng.random() < 0.6 else int(rng.integers(0, 1 << bw))
    nonnull = O.fle_pred(defs, n, def_bw, O.OP_EQ, max_def)
    sub = O.fle_pred(enc, n_data, bw, op, c) if n_data else np.zeros(1, np.uint64)
    if n_data < k:   # data rows that do not exist select nothing
        sb = np.zeros(max(k, 1), bool)
        sb[:n_data] = bits_of(sub, n_data)
        sub = np.packbits(np.concatenate([sb, np.zeros((-len(sb)) % 64, bool)]), bitorder="little").view(np.uint64)
    exp = O.bitmap_expand(nonnull, sub, n)
    out = torch.full(((n + 63) // 64 + 1,), -1, dtype=torch.int64, device="cuda")
    capi.fle_pred_nullable(d_defs, def_bw, max_def, n, d_enc, n_data, bw, op, c, bitmap=out)
    got = words(out)[:(n + 63) // 64]
    if not np.array_equal(got, exp):
        bad += 1
        print("leaf mismatch", dict(bw=bw, n=n, max_def=max_def, k=k, n_data=n_data, op=op), flush=True)
        continue
    # late materialisation of the rows the leaf selected, or of a random selection
    if rng.random() < 0.5:
        selbits = bits_of(exp, n)
        d_sel = out[:(n + 63) // 64 + 1]
    else:
        selbits = rng.random(n) < rng.random()
        d_sel = dev(np.packbits(np.concatenate([selbits, np.zeros((-n) % 64, bool)]), bitorder="little").view(np.uint64))
    if bw <= 16 and rng.random() < 0.5:
        D = 1 << bw
        entries = (np.arange(D, dtype=np.int64) * 7 - 1000)
        if rng.random() < 0.5:
            dd, ent = capi.Dict(entries.astype(np.int32).view(np.uint8), capi.T_INT32), entries.astype(np.int32).astype(np.int64)
        else:
            entries = entries * (1 << 33)
            dd, ent = capi.Dict(entries.view(np.uint8), capi.T_INT64), entries
        values = ent[vals] if k else np.zeros(0, np.int64)
    else:
        dd, values = None, vals.astype(np.int64)
    rank = np.cumsum(is_set) - 1
    take = selbits & is_set & (rank < min(n_data, k))
    exp_dense = values[rank[take]]
    dense, flags, n_sel, n_val = capi.select_nullable(dd, d_defs, def_bw, max_def, n, d_enc, n_data, bw, d_sel)
    got = dense.cpu().numpy()
    got = got.view(np.uint32).astype(np.int64) if dd is None else got.astype(np.int64)
    if (n_sel != int(selbits.sum()) or n_val != len(exp_dense) or not np.array_equal(got, exp_dense)
            or not np.array_equal(bits_of(words(flags), n_sel), is_set[selbits])):
        bad += 1
        print("select_nullable mismatch", dict(bw=bw, n=n, max_def=max_def, k=k, n_data=n_data, dict=dd is not None), flush=True)
    if dd is not None:
        dd.close()
    if op != 5 and n_data == k:
        # the same leaf and-ed / or-ed into a REQUIRED column's predicate through the program
        other = rng.integers(0, 8, n).astype(np.uint32)
        d_other = dev(O.fle_encode(other, 3))
        cols = [capi.fle_column(d_other, 3), capi.nullable_fle_column(d_defs, def_bw, max_def, d_enc, bw, n_data)]
        leafbits = bits_of(exp, n)
        L, AND, OR = capi.leaf, capi.and_node, capi.or_node
        for node, truth in ((AND(), (other < 5) & leafbits), (OR(), (other < 5) | leafbits)):
            g = bits_of(words(capi.eval_program([L(0, O.OP_LT, 5), L(1, op, c), node], cols, n)), n)
            if not np.array_equal(g, truth):
                bad += 1
                print("program mismatch", dict(bw=bw, n=n, max_def=max_def, op=op, node=node), flush=True)
print("nullable soak done, mismatches:", bad)
sys.exit(1 if bad else 0)

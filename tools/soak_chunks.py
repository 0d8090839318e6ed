#!/usr/bin/env python3
"""dev tool: randomized differential soak of the page-list calls (ips_chunk_*): random page cuts
(tiny pages, empty pages, whole sub-tiles, odd sizes), random widths incl. runs of different widths,
REQUIRED / OPTIONAL / PLAIN columns, random AND / OR trees -- against the oracle page by page."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as entry  # noqa: E402

ips = entry.load_package()
capi = ips.capi
O = entry.load_oracle()
rng = np.random.default_rng(int(os.environ.get("IPS_SOAK_SEED", "20261005")))
ITERS = int(os.environ.get("IPS_SOAK_ITERS", "60"))


def words(t):
    return t.cpu().numpy().view(np.uint64)


def dev(a):
    a = np.ascontiguousarray(a)
    return torch.from_numpy(a.view(np.int64).copy()).cuda() if a.size else torch.zeros(2, dtype=torch.int64, device="cuda")


def bits(w, n):
    return np.unpackbits(np.ascontiguousarray(w).view(np.uint8), bitorder="little")[:n].astype(bool)


def pack(b):
    n = len(b)
    x = np.zeros(((n + 63) // 64) * 64, np.uint8)
    x[:n] = b
    return np.packbits(x, bitorder="little").view(np.uint64)


def cuts(n):
    style = rng.integers(0, 4)
    pool = [[0, 1, 2, 31, 32, 33, 63, 64, 65], [2048, 4096, 2048 * 5], [100, 2047, 2049, 70001, 5000],
            [int(rng.integers(1, 40000))]][style]
    out, left = [], n
    while left > 0:
        m = min(int(rng.choice(pool)), left)
        out.append(m)
        left -= m
    return out


def fle_col(n, optional):
    """-> (chunk, expect(op, consts) -> bits, values, is_set)"""
    base_w = int(rng.integers(1, 33)) if not optional else int(rng.integers(1, 17))
    page_rows = cuts(n)
    is_set = rng.random(n) >= rng.choice([0.0, 0.1, 0.6]) if optional else np.ones(n, bool)
    pages, host, pos = [], [], 0
    vals = np.zeros(n, np.uint32)
    for m in page_rows:
        if m == 0:
            pages.append((None, 0, base_w, torch.zeros(2, dtype=torch.int64, device="cuda"), 0) if optional else (None, 0, base_w))
            continue
        w = base_w if rng.random() < 0.8 else int(rng.integers(1, base_w + 1))   # a run of another width
        v = rng.integers(0, 1 << w, m, dtype=np.uint64).astype(np.uint32)
        vals[pos:pos + m] = v
        s = is_set[pos:pos + m]
        if optional:
            k = int(s.sum())
            defs = O.fle_encode(s.astype(np.uint32), 1)
            enc = O.fle_encode(v[s], w) if k else np.zeros(2, np.uint64)
            pages.append((dev(enc), m, w, dev(defs), ((k + 63) // 64) * 64))
            host.append((enc, m, w, defs, k))
        else:
            enc = O.fle_encode(v, w)
            pages.append((dev(enc), m, w))
            host.append((enc, m, w, None, m))
        pos += m
    chunk = capi.Chunk(pages, max_def_level=1 if optional else 0)

    def expect(op, consts):
        out = []
        for enc, m, w, defs, k in host:
            lim = (1 << w) - 1
            cs = [int(c) for c in np.atleast_1d(consts)]
            if op == O.OP_IN:
                fit = [c for c in cs if c <= lim]
                sub = O.fle_pred(enc, k, w, op, fit) if (fit and k) else np.zeros(max((k + 63) // 64, 1), np.uint64)
            elif cs[0] > lim:
                sub = pack(np.full(k, op in (O.OP_LT, O.OP_LE)))
            else:
                sub = O.fle_pred(enc, k, w, op, cs[0]) if k else np.zeros(1, np.uint64)
            if defs is None:
                out.append(bits(sub, m))
            else:
                nonnull = O.fle_pred(defs, m, 1, O.OP_EQ, 1)
                out.append(bits(O.bitmap_expand(nonnull, sub, m), m))
        return np.concatenate(out) if out else np.zeros(0, bool)
    return chunk, expect, vals, is_set, base_w


def plain_col(n):
    t = int(rng.choice([capi.T_INT32, capi.T_INT64, capi.T_FLOAT, capi.T_DOUBLE]))
    npt = capi.NP_TYPES[t]
    vals = (rng.normal(0, 100, n) if t in (capi.T_FLOAT, capi.T_DOUBLE) else rng.integers(-500, 500, n)).astype(npt)
    pages, host, pos = [], [], 0
    for m in cuts(n):
        d = torch.from_numpy(np.ascontiguousarray(vals[pos:pos + m])).cuda() if m else torch.zeros(4, dtype=torch.int64, device="cuda")
        pages.append((d, m, 0))
        if m:
            host.append((O.plain_encode(vals[pos:pos + m], t), m))
        pos += m
    chunk = capi.Chunk(pages, encoding=capi.COL_PLAIN, type_=t)

    def expect(op, lit):
        return np.concatenate([bits(O.plain_pred(pg, m, t, op, lit, O.SEM_SQL), m) for pg, m in host])
    return chunk, expect, vals, t


bad = 0
optional_sets = {}
for it in range(ITERS):
    n = int(rng.choice([1, 70, 3000, 2048 * 7, int(rng.integers(1, 250000))]))
    cols = []
    for c in range(int(rng.integers(1, 4))):
        kind = rng.integers(0, 3)
        if kind == 2:
            ch, ex, vals, t = plain_col(n)
            cols.append(("plain", ch, ex, vals, t))
        else:
            ch, ex, vals, is_set, w = fle_col(n, kind == 1)
            cols.append(("fle", ch, ex, vals, w))
            if kind == 1:
                optional_sets[id(ch)] = is_set
    leaves = []
    for _ in range(int(rng.integers(1, 6))):
        ci = int(rng.integers(0, len(cols)))
        kind, ch, ex, vals, extra = cols[ci]
        if kind == "plain":
            op = int(rng.integers(0, 5))
            lit = vals[rng.integers(0, n)]
            leaves.append((capi.plain_leaf(ci, op, lit, extra), ex(op, lit)))
        else:
            op = int(rng.integers(0, 6))
            if op == 5:
                cs = [int(x) for x in rng.choice(vals, int(rng.integers(1, 12)))]
            else:
                cs = int(vals[rng.integers(0, n)]) if rng.random() < 0.7 else int(rng.integers(0, 1 << extra))
            leaves.append((capi.leaf(ci, op, cs), ex(op, cs)))
    nodes, stack = [], []
    for node, e in leaves:
        nodes.append(node)
        stack.append(e)
        while len(stack) > 1 and rng.random() < 0.6:
            b, a = stack.pop(), stack.pop()
            if rng.random() < 0.5:
                nodes.append(capi.and_node()); stack.append(a & b)
            else:
                nodes.append(capi.or_node()); stack.append(a | b)
    while len(stack) > 1:
        b, a = stack.pop(), stack.pop()
        nodes.append(capi.and_node()); stack.append(a & b)
    got = capi.eval_program_chunks(nodes, [c[1] for c in cols])
    if not np.array_equal(words(got), pack(stack[0])):
        bad += 1
        print("program mismatch: iter", it, "n", n, "nodes", len(nodes), [c[0] for c in cols])
    # a fused scan on the first REQUIRED FLE column, if any
    for kind, ch, ex, vals, extra in cols:
        if kind == "fle" and ch.n_rows == n and not any(len(p) > 3 for p in ch.keep):
            op = int(rng.integers(1, 5))
            c = int(vals[rng.integers(0, n)])
            bm, bv, cnt = ch.fle_scan(op, c)
            e = ex(op, c)
            if not np.array_equal(words(bm), pack(e)):
                bad += 1; print("scan bitmap mismatch", it, n)
            elif not np.array_equal(ch.compact(bv, cnt).cpu().numpy().view(np.uint32), vals[e]):
                bad += 1; print("scan values mismatch", it, n)
            break
    # late materialisation of the OPTIONAL FLE columns across their pages (ips_chunk_select_nullable)
    for kind, ch, ex, vals, extra in cols:
        if kind == "fle" and id(ch) in optional_sets:
            is_set = optional_sets[id(ch)]
            sel = rng.random(n) < float(rng.choice([0.02, 0.3, 1.0]))
            bm = torch.from_numpy(np.concatenate([pack(sel), np.zeros(2, np.uint64)]).view(np.int64)).cuda()
            dense, flags, n_sel, n_val, bad_idx = ch.select_nullable(bm)
            if n_sel != int(sel.sum()) or n_val != int((sel & is_set).sum()) or bad_idx:
                bad += 1; print("select_nullable counts mismatch", it, n)
            elif not np.array_equal(dense.cpu().numpy().view(np.uint32), vals[sel & is_set]):
                bad += 1; print("select_nullable values mismatch", it, n)
            elif not np.array_equal(bits(words(flags), n_sel), is_set[sel]):
                bad += 1; print("select_nullable flags mismatch", it, n)
    optional_sets.clear()
    for c in cols:
        c[1].close()
print("soak done, mismatches:", bad)

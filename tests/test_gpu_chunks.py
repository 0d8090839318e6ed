"""Column chunks as LISTS OF PAGES (ips_chunk_*): the page loop of the scanner -- ReadDataPage /
InitDataPage per page, batches cut at every column's page end (hdfs-parquet-scanner.cc:730-924,
1837-1855) -- inside one launch per run of equally wide pages.  Expected results are the oracle's,
page by page, concatenated in row order: what the reference's per-batch bitsets add up to."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev_words(a):
    a = np.ascontiguousarray(a)
    if a.size == 0:
        return torch.zeros(2, dtype=torch.int64, device="cuda")
    return torch.from_numpy(a.view(np.int64).copy()).cuda()


def words(t):
    return t.cpu().numpy().view(np.uint64)


def bits_of(w, n):
    return np.unpackbits(np.ascontiguousarray(w).view(np.uint8), bitorder="little")[:n].astype(bool)


def pack(bits):
    n = len(bits)
    b = np.zeros(((n + 63) // 64) * 64, dtype=np.uint8)
    b[:n] = bits
    return np.packbits(b, bitorder="little").view(np.uint64)


def cuts(rng, n, sizes):
    """page row counts that add up to n, drawn from `sizes` (0 = an empty page)"""
    out, left = [], n
    while left > 0:
        s = int(rng.choice(sizes))
        s = min(s, left)
        out.append(s)
        left -= s
    return out


RAGGED = [0, 1, 5, 31, 32, 33, 63, 64, 65, 100, 2047, 2048, 2049, 4097, 10000, 70001]


def fle_chunk(capi, O, vals, page_rows, bw_of_page):
    """-> (Chunk, [(enc words, n, bw)]) for a REQUIRED FLE column cut into pages"""
    pages, host, pos = [], [], 0
    for i, m in enumerate(page_rows):
        bw = bw_of_page(i)
        enc = O.fle_encode(vals[pos:pos + m], bw) if m else np.zeros(2, np.uint64)
        pages.append((dev_words(enc), m, bw))
        host.append((enc, m, bw))
        pos += m
    return capi.Chunk(pages), host


def oracle_fle_pred_pages(O, host, op, consts):
    """the reference's answer page by page (constants that do not fit a page's width: unsigned SQL meaning)"""
    out = []
    for enc, m, bw in host:
        if m == 0:
            continue
        lim = (1 << bw) - 1
        cs = np.atleast_1d(consts)
        if op == O.OP_IN:
            fit = [int(c) for c in cs if c <= lim]
            out.append(bits_of(O.fle_pred(enc, m, bw, op, fit), m) if fit else np.zeros(m, bool))
        elif cs[0] > lim:
            out.append(np.full(m, op in (O.OP_LT, O.OP_LE)))
        else:
            out.append(bits_of(O.fle_pred(enc, m, bw, op, int(cs[0])), m))
    return np.concatenate(out) if out else np.zeros(0, bool)


@pytest.mark.parametrize("seed", range(6))
def test_chunk_fle_scan_ragged_pages(capi, O, seed):
    """ips_chunk_fle_scan / ips_chunk_select over pages of every awkward size: bitmap = the pages'
    oracle bitmaps concatenated at their row offsets (unaligned dwords, shared words between pages,
    empty pages), values = FleDecoder::Get(val, skip) of the selected rows, page by page."""
    rng = np.random.default_rng(100 + seed)
    bw = int(rng.choice([1, 3, 8, 12, 16, 21, 32]))
    n = int(rng.choice([1, 77, 5000, 200003, 700001]))
    page_rows = cuts(rng, n, RAGGED)
    vals = rng.integers(0, 1 << bw, n, dtype=np.uint64).astype(np.uint32)
    chunk, host = fle_chunk(capi, O, vals, page_rows, lambda i: bw)
    assert chunk.n_rows == n
    c = int(vals[n // 2])
    for op, consts in ((O.OP_LT, c), (O.OP_GE, c), (O.OP_EQ, c), (O.OP_IN, [c, int(vals[0]), 0])):
        outs = chunk.alloc_outputs()
        outs[0].fill_(-1)                                   # the bitmap is written, never OR-ed into
        bitmap, bvals, counts = chunk.fle_scan(op, consts, outputs=outs)
        exp = oracle_fle_pred_pages(O, host, op, consts)
        assert np.array_equal(words(bitmap), pack(exp)), (seed, bw, n, op)
        exp_vals = np.concatenate([O.fle_select(enc, m, b, pack(exp[p0:p0 + m])) for (enc, m, b), p0 in
                                   zip(host, np.cumsum([0] + page_rows[:-1])) if m] or [np.zeros(0, np.uint32)])
        dense = chunk.compact(bvals, counts).cpu().numpy().view(np.uint32)
        assert np.array_equal(dense, exp_vals), (seed, bw, n, op)
        assert np.array_equal(exp_vals, vals[exp])
        # late materialisation against the chunk-wide bitmap reproduces the fused batches
        bv2, cnt2 = chunk.select(bitmap)
        assert torch.equal(cnt2, counts)
        assert np.array_equal(chunk.compact(bv2, cnt2).cpu().numpy().view(np.uint32), exp_vals)
    chunk.close()


def test_chunk_pages_of_growing_width(capi, O):
    """The dictionary writer stores each data page's code width in its first byte and the width
    grows with the dictionary (dict-encoding.h:425-447, quirk Q8): runs of pages of different widths
    in one chunk; codes / constants that do not fit an early page's width."""
    rng = np.random.default_rng(7)
    D = 3000
    dict_vals = np.sort(rng.choice(np.arange(-10 ** 6, 10 ** 6), D, replace=False)).astype(np.int32)
    page_rows = [3000, 64, 70001, 1, 2048, 33333, 5]
    widths = [4, 4, 7, 7, 9, 12, 12]
    codes = np.concatenate([rng.integers(0, min(D, 1 << w), m) for m, w in zip(page_rows, widths)]).astype(np.uint32)
    chunk, host = fle_chunk(capi, O, codes, page_rows, lambda i: widths[i])
    dd = capi.Dict(dict_vals.view(np.uint8), capi.T_INT32)
    col = dict_vals[codes]
    n = len(codes)
    for op, lit in ((capi.OP_LT, dict_vals[100]), (capi.OP_GE, dict_vals[700]), (capi.OP_EQ, dict_vals[2999]),
                    (capi.OP_EQ, dict_vals[3]), (capi.OP_LE, dict_vals[0] - 1), (capi.OP_GT, dict_vals[0] - 1),
                    (capi.OP_IN, [dict_vals[1], dict_vals[90], dict_vals[2000], 12345678])):
        bitmap, bvals, counts = chunk.dict_scan(dd, op, lit)
        truth = {capi.OP_LT: lambda: col < lit, capi.OP_GE: lambda: col >= lit, capi.OP_EQ: lambda: col == lit,
                 capi.OP_LE: lambda: col <= lit, capi.OP_GT: lambda: col > lit,
                 capi.OP_IN: lambda: np.isin(col, lit)}[op]()
        kind, fle_op, cs = dd.translate(op, lit)
        if kind == capi.XL_FLE:                              # the oracle on the codes, page by page
            assert np.array_equal(truth, oracle_fle_pred_pages(O, host, fle_op, cs))
        assert np.array_equal(bits_of(words(bitmap), n), truth), op
        assert np.array_equal(chunk.compact(bvals, counts).cpu().numpy(), col[truth]), op
        bv2, cnt2 = chunk.select(bitmap, dd)
        assert np.array_equal(chunk.compact(bv2, cnt2).cpu().numpy(), col[truth]), op
    dd.close()
    chunk.close()


def oracle_nullable_page(O, defs, n, enc, k, bw, op, consts):
    nonnull = O.fle_pred(defs, n, 1, O.OP_EQ, 1)
    sub = O.fle_pred(enc, k, bw, op, consts) if k else np.zeros(1, np.uint64)
    return bits_of(O.bitmap_expand(nonnull, sub, n), n)


@pytest.mark.parametrize("seed", range(4))
def test_chunk_program_pages_unaligned_between_columns(capi, O, seed):
    """ips_eval_program_chunks: five columns of one row group whose pages end at different rows
    (REQUIRED FLE, REQUIRED dictionary codes, OPTIONAL FLE, PLAIN int32, PLAIN int64), AND / OR trees,
    BETWEEN pairs, IN lists.  Every leaf's expected bits come from the oracle page by page
    (fle-encoding.h:7962-8313; nullable: hdfs-parquet-scanner.cc:326-345; PLAIN: parquet-common.h:197-250)."""
    rng = np.random.default_rng(500 + seed)
    n = int(rng.choice([9, 4100, 150001, 600007]))
    sizes = [RAGGED, [1000, 4096, 65536], [2048 * 3, 777], [64, 128, 100000]][seed]
    # column 0: REQUIRED FLE w=12; column 1: REQUIRED w=5
    v0 = rng.integers(0, 1 << 12, n).astype(np.uint32)
    v1 = rng.integers(0, 1 << 5, n).astype(np.uint32)
    ch0, h0 = fle_chunk(capi, O, v0, cuts(rng, n, sizes), lambda i: 12)
    ch1, h1 = fle_chunk(capi, O, v1, cuts(rng, n, RAGGED), lambda i: 5)
    # column 2: OPTIONAL FLE w=7, ~20 % NULL
    is_set = rng.random(n) >= 0.2
    v2 = rng.integers(0, 1 << 7, n).astype(np.uint32)          # value of row r if not NULL
    pages2, host2, pos = [], [], 0
    for m in cuts(rng, n, sizes):
        if m == 0:
            pages2.append((None, 0, 7, torch.zeros(2, dtype=torch.int64, device="cuda"), 0))
            continue
        s = is_set[pos:pos + m]
        k = int(s.sum())
        defs = O.fle_encode(s.astype(np.uint32), 1)
        enc = O.fle_encode(v2[pos:pos + m][s], 7) if k else np.zeros(2, np.uint64)
        pages2.append((dev_words(enc), m, 7, dev_words(defs), ((k + 63) // 64) * 64))
        host2.append((defs, m, enc, k))
        pos += m
    ch2 = capi.Chunk(pages2, max_def_level=1)
    # columns 3 / 4: PLAIN int32 / int64
    p32 = rng.integers(-1000, 1000, n).astype(np.int32)
    p64 = rng.integers(-10 ** 12, 10 ** 12, n).astype(np.int64)

    def plain_chunk(vals, t):
        pages, host, pos = [], [], 0
        for m in cuts(rng, n, RAGGED if seed == 0 else sizes):
            page = O.plain_encode(vals[pos:pos + m], t)
            d = torch.from_numpy(np.ascontiguousarray(vals[pos:pos + m])).cuda() if m else torch.zeros(4, dtype=torch.int64, device="cuda")
            pages.append((d, m, 0))
            host.append((page, m))
            pos += m
        return capi.Chunk(pages, encoding=capi.COL_PLAIN, type_=t), host
    ch3, h3 = plain_chunk(p32, capi.T_INT32)
    ch4, h4 = plain_chunk(p64, capi.T_INT64)
    chunks = [ch0, ch1, ch2, ch3, ch4]

    def fle_leaf(col, host, op, consts):
        return capi.leaf(col, op, consts), oracle_fle_pred_pages(O, host, op, consts)

    def null_leaf(op, consts):
        exp = np.concatenate([oracle_nullable_page(O, d, m, e, k, 7, op, consts) for d, m, e, k in host2])
        return capi.leaf(2, op, consts), exp

    def plain_leaf(col, host, t, op, lit):
        exp = np.concatenate([bits_of(O.plain_pred(pg, m, t, op, lit, O.SEM_SQL), m) for pg, m in host if m]
                             or [np.zeros(0, bool)])
        return capi.plain_leaf(col, op, lit, t), exp

    AND, OR = capi.and_node, capi.or_node
    a = fle_leaf(0, h0, O.OP_GE, 1000)
    b = fle_leaf(0, h0, O.OP_LE, 3000)
    c = fle_leaf(1, h1, O.OP_IN, [3, 17, 30])
    d = null_leaf(O.OP_LT, 40)
    e = null_leaf(O.OP_GE, 10)
    f = plain_leaf(3, h3, capi.T_INT32, O.OP_GT, np.int32(-250))
    g = plain_leaf(4, h4, capi.T_INT64, O.OP_LT, np.int64(3 * 10 ** 11))
    h = plain_leaf(4, h4, capi.T_INT64, O.OP_GE, np.int64(-5 * 10 ** 11))
    trees = [
        ([a[0]], a[1]),
        ([d[0]], d[1]),
        ([g[0]], g[1]),
        ([a[0], b[0], AND()], a[1] & b[1]),                                       # BETWEEN: one pass
        ([a[0], b[0], AND(), c[0], AND(), d[0], AND()], a[1] & b[1] & c[1] & d[1]),
        ([d[0], e[0], AND(), f[0], OR()], (d[1] & e[1]) | f[1]),                  # pair on the OPTIONAL column
        ([g[0], h[0], AND(), c[0], OR(), a[0], AND()], ((g[1] & h[1]) | c[1]) & a[1]),
        ([a[0], c[0], AND(), d[0], f[0], AND(), OR(), g[0], AND()], ((a[1] & c[1]) | (d[1] & f[1])) & g[1]),
    ]
    for nodes, exp in trees:
        got = capi.eval_program_chunks(nodes, chunks)
        assert np.array_equal(words(got), pack(exp)), (seed, n, len(nodes))
    # row-model cross-check of one tree on the raw values
    truth = (v0 >= 1000) & (v0 <= 3000) & np.isin(v1, [3, 17, 30]) & is_set & (v2 < 40)
    assert np.array_equal(trees[4][1], truth)
    for ch in chunks:
        ch.close()


def test_chunk_plain_scan_pages(capi, O):
    """ips_chunk_plain_scan over ragged PLAIN pages (int32: a dword per lane; int64: a dword per lane
    pair), both operand orders, BETWEEN pair: bitmap vs the oracle page by page, slots vs the rows."""
    rng = np.random.default_rng(11)
    n = 123457
    for t, vals in ((capi.T_INT32, rng.integers(-1000, 1000, n).astype(np.int32)),
                    (capi.T_INT64, rng.integers(-10 ** 9, 10 ** 9, n).astype(np.int64)),
                    (capi.T_DOUBLE, rng.normal(0, 100, n))):
        page_rows = cuts(rng, n, RAGGED)
        pages, host, pos = [], [], 0
        for m in page_rows:
            d = torch.from_numpy(np.ascontiguousarray(vals[pos:pos + m])).cuda() if m else torch.zeros(4, dtype=torch.int64, device="cuda")
            pages.append((d, m, 0))
            host.append((O.plain_encode(vals[pos:pos + m], t), m))
            pos += m
        chunk = capi.Chunk(pages, encoding=capi.COL_PLAIN, type_=t)
        lit = vals[n // 3]
        for sem in (O.SEM_REFERENCE, O.SEM_SQL):
            exp = np.concatenate([bits_of(O.plain_pred(pg, m, t, O.OP_LT, lit, sem), m) for pg, m in host if m])
            bitmap, bvals, counts = chunk.plain_scan(capi.OP_LT, lit, semantics=sem)
            assert np.array_equal(words(bitmap), pack(exp)), (t, sem)
            got = chunk.compact(bvals, counts).cpu().numpy()
            assert np.array_equal(got.view(vals.dtype), vals[exp]), (t, sem)
        lo, hi = np.sort(vals[:2])
        bitmap, bvals, counts = chunk.plain_scan(capi.OP_GE, lo, op2=capi.OP_LE, literal2=hi)
        exp = (vals >= lo) & (vals <= hi)
        assert np.array_equal(words(bitmap), pack(exp))
        assert np.array_equal(chunk.compact(bvals, counts).cpu().numpy().view(vals.dtype), vals[exp])
        # ReadValue(skip) against selections over the chunk's rows (ips_chunk_select on PLAIN pages)
        for sel in (exp, rng.random(n) < 0.02, rng.random(n) < 0.7, np.ones(n, bool)):
            bm = torch.from_numpy(pack(sel).view(np.int64).copy()).cuda()
            bv, cnt = chunk.select(bm)
            assert np.array_equal(chunk.compact(bv, cnt).cpu().numpy().view(vals.dtype), vals[sel]), t
        chunk.close()


def test_chunk_aligned_pages_equal_contiguous(capi, ips, O):
    """256 whole pages of 2^16 rows per column = the contiguous column: the paged program and scan
    produce the very words of ips_eval_program / ips_dict_scan on one buffer."""
    n_pages, pr = 64, 1 << 16
    n = n_pages * pr
    q6 = ips.q6
    codes = [q6.codes_gpu(capi, c, n) for c in range(3)]
    encs = [capi.fle_encode(codes[c], q6.COLUMNS[c][3]) for c in range(3)]
    nodes, cols = q6.program(capi, encs)
    ref = capi.eval_program(nodes, cols, n)
    chunks = []
    for c in range(3):
        w = q6.COLUMNS[c][3]
        wpp = pr // 64 * w
        chunks.append(capi.Chunk([(encs[c][p * wpp:(p + 1) * wpp].clone(), pr, w) for p in range(n_pages)]))
    got = capi.eval_program_chunks(nodes, chunks)
    assert torch.equal(got, ref)
    for ch in chunks:
        ch.close()


@pytest.mark.parametrize("seed", range(3))
def test_chunk_tiny_pages_share_words(capi, O, seed):
    """Hundreds of pages of 1..70 rows: several pages inside one bitmap dword, every kernel family
    (FLE predicate incl. w = 32 early pruning, nullable leaf, PLAIN 4- and 8-byte) in store / AND / OR
    mode against the oracle page by page."""
    rng = np.random.default_rng(900 + seed)
    n = 9000 + int(rng.integers(0, 64))
    tiny = list(range(0, 71))
    v0 = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    v0[::5] = (v0[7] & 0xFFFF0000) | (v0[::5] & 0xFFFF)        # rows undecided after the high planes
    ch0, h0 = fle_chunk(capi, O, v0, cuts(rng, n, tiny), lambda i: 32)
    v1 = rng.integers(0, 1 << 9, n).astype(np.uint32)
    ch1, h1 = fle_chunk(capi, O, v1, cuts(rng, n, tiny), lambda i: 9)
    is_set = rng.random(n) >= 0.3
    v2 = rng.integers(0, 1 << 4, n).astype(np.uint32)
    pages2, host2, pos = [], [], 0
    for m in cuts(rng, n, [m for m in tiny if m]):
        s_ = is_set[pos:pos + m]
        k = int(s_.sum())
        defs = O.fle_encode(s_.astype(np.uint32), 1)
        enc = O.fle_encode(v2[pos:pos + m][s_], 4) if k else np.zeros(2, np.uint64)
        pages2.append((dev_words(enc), m, 4, dev_words(defs), ((k + 63) // 64) * 64))
        host2.append((defs, m, enc, k))
        pos += m
    ch2 = capi.Chunk(pages2, max_def_level=1)
    p64 = rng.integers(-10 ** 6, 10 ** 6, n).astype(np.int64)
    pages3, host3, pos = [], [], 0
    for m in cuts(rng, n, [m for m in tiny if m]):
        pages3.append((torch.from_numpy(np.ascontiguousarray(p64[pos:pos + m])).cuda(), m, 0))
        host3.append((O.plain_encode(p64[pos:pos + m], capi.T_INT64), m))
        pos += m
    ch3 = capi.Chunk(pages3, encoding=capi.COL_PLAIN, type_=capi.T_INT64)
    c0 = int(v0[7])
    a = (capi.leaf(0, O.OP_LT, c0), oracle_fle_pred_pages(O, h0, O.OP_LT, c0))
    b = (capi.leaf(0, O.OP_GE, c0 & 0xFFFF0000), oracle_fle_pred_pages(O, h0, O.OP_GE, c0 & 0xFFFF0000))
    c = (capi.leaf(1, O.OP_IN, [1, 100, 511]), oracle_fle_pred_pages(O, h1, O.OP_IN, [1, 100, 511]))
    d = (capi.leaf(2, O.OP_GE, 7), np.concatenate([oracle_nullable_page(O, df, m, e, k, 4, O.OP_GE, 7) for df, m, e, k in host2]))
    e = (capi.plain_leaf(3, O.OP_LT, np.int64(1000), capi.T_INT64),
         np.concatenate([bits_of(O.plain_pred(pg, m, capi.T_INT64, O.OP_LT, np.int64(1000), O.SEM_SQL), m) for pg, m in host3]))
    AND, OR = capi.and_node, capi.or_node
    chunks = [ch0, ch1, ch2, ch3]
    for nodes, exp in (([a[0]], a[1]), ([d[0]], d[1]), ([e[0]], e[1]),
                       ([a[0], b[0], AND()], a[1] & b[1]),
                       ([c[0], d[0], AND(), e[0], AND(), a[0], OR()], (c[1] & d[1] & e[1]) | a[1]),
                       ([e[0], d[0], OR(), c[0], OR(), b[0], AND()], (e[1] | d[1] | c[1]) & b[1])):
        got = capi.eval_program_chunks(nodes, chunks)
        assert np.array_equal(words(got), pack(exp)), (seed, len(nodes))
    bitmap, bvals, counts = ch1.fle_scan(O.OP_IN, [1, 100, 511])
    assert np.array_equal(words(bitmap), pack(c[1]))
    assert np.array_equal(ch1.compact(bvals, counts).cpu().numpy().view(np.uint32), v1[c[1]])
    for ch in chunks:
        ch.close()


def test_bitmap_batch_counts(capi):
    rng = np.random.default_rng(5)
    for n in (1, 2047, 2048, 2049, 100003):
        bits = rng.random(n) < 0.3
        bm = torch.from_numpy(pack(bits).view(np.int64).copy()).cuda()
        got = capi.bitmap_batch_counts(bm, n).cpu().numpy()
        exp = np.add.reduceat(bits.astype(np.int64), np.arange(0, n, 2048))
        assert np.array_equal(got, exp), n


@pytest.mark.parametrize("common", [True, False])
@pytest.mark.parametrize("seed", range(5))
def test_chunk_program_one_pass_over_common_pages(capi, O, seed, common, request):
    """ips_eval_program_chunks on a conjunct / disjunct chain over REQUIRED FLE chunks whose pages hold the SAME
    rows in every chunk (what a writer that flushes all columns of a row group together produces): ONE launch,
    blockIdx.y = page, every operand in its own block geometry (ips_chain.hip).  Pages of every awkward size --
    page starts inside bitmap dwords (edge slots + fix-up), pages smaller than a dword, empty pages -- against
    numpy on the raw values and against the per-operand plan; constants that do not fit a chunk's width.
    common = False: every column is cut at its OWN rows (what column writers produce): the chain walks the segments
    between neighbouring page starts of any operand (blockIdx.y = segment), each operand from wherever the segment
    starts inside its page."""
    request.addfinalizer(lambda: capi.set_program_strategy(capi.PROGRAM_AUTO))
    rng = np.random.default_rng(900 + seed)
    n = int(rng.choice([1, 37, 4096, 70001, 300007]))
    sizes = [RAGGED, [2048, 4096], [1, 5, 31, 33], [70001, 10000, 0], [32, 64, 2048 + 32]][seed]
    page_rows = cuts(rng, n, sizes)
    n_ops = int(rng.integers(2, 6))
    L, AND, OR = capi.leaf, capi.and_node, capi.or_node
    cmp_np = {O.OP_EQ: np.equal, O.OP_LT: np.less, O.OP_LE: np.less_equal, O.OP_GT: np.greater, O.OP_GE: np.greater_equal}
    chunks, nodes, exp = [], [], None
    for i in range(n_ops):
        w = int(rng.integers(1, 25))
        span = (1 << w) - 1 if rng.random() < 0.6 else min((1 << w) - 1, 15)
        v = rng.integers(0, span + 1, n, dtype=np.uint64).astype(np.uint32)
        own = page_rows if common else cuts(rng, n, [RAGGED, [2048, 4096, 100], [1, 5, 31, 33, 700], [70001, 10000, 0], [32, 64, 2048 + 32, 1984]][(seed + i) % 5])
        ch, _ = fle_chunk(capi, O, v, own, lambda k, w=w: w)
        chunks.append(ch)
        kind = rng.integers(0, 3)
        # now and then a constant beyond the column's width (unsigned SQL meaning: LT / LE always, else never)
        big = rng.random() < 0.15 and w < 24
        if kind == 0:
            op = int(rng.integers(0, 5))
            c = (1 << w) + 3 if big else int(rng.integers(0, span + 1))
            nodes.append(L(i, op, c))
            sel = cmp_np[op](v.astype(np.uint64), np.uint64(c))
        elif kind == 1:
            op1, op2 = int(rng.integers(0, 5)), int(rng.integers(0, 5))
            c1 = int(rng.integers(0, span + 1))
            c2 = (1 << w) + 1 if big else int(rng.integers(0, span + 1))
            nodes += [L(i, op1, c1), L(i, op2, c2)]
            s1, s2 = cmp_np[op1](v.astype(np.uint64), np.uint64(c1)), cmp_np[op2](v.astype(np.uint64), np.uint64(c2))
            if rng.random() < 0.7:
                nodes.append(AND())
                sel = s1 & s2
            else:
                nodes.append(OR())
                sel = s1 | s2
        else:
            members = [int(x) for x in rng.integers(0, span + 1, int(rng.integers(1, 17)))]
            if big:
                members.append((1 << w) + 7)
            nodes.append(L(i, O.OP_IN, members))
            sel = np.isin(v, np.array(members, dtype=np.uint64))
        if i == 0:
            exp = sel
        elif rng.random() < 0.75:
            nodes.append(AND())
            exp = exp & sel
        else:
            nodes.append(OR())
            exp = exp | sel
    # (ONE_PASS: chunks cut at different rows take the segmented chain; AUTO keeps the per-operand launches for them)
    for strat in (capi.PROGRAM_AUTO, capi.PROGRAM_ONE_PASS, capi.PROGRAM_PER_OPERAND):
        capi.set_program_strategy(strat)
        bm = torch.full(((n + 63) // 64 + 2,), -1, dtype=torch.int64, device="cuda")  # (no zero-initialised bitmap needed)
        capi.eval_program_chunks(nodes, chunks, bitmap=bm)
        got_words = words(bm)[:(n + 63) // 64]
        assert np.array_equal(bits_of(got_words, n), exp), (seed, common, strat, n, page_rows[:8])
        if n % 64:
            assert int(got_words[-1]) >> (n % 64) == 0, ("bits behind the last row are zero", seed, common, strat, page_rows[-4:])
    for ch in chunks:
        ch.close()


@pytest.mark.parametrize("seed", range(6))
def test_chunk_select_nullable_pages(capi, O, seed):
    """ips_chunk_select_nullable: late materialisation of an OPTIONAL column across its pages
    (ReadValue(skip) + ReadDefinitionLevel over ReadDataPage boundaries, hdfs-parquet-scanner.cc:927-979, 1006-1038):
    dense values of the selected NOT-NULL rows in row order, one NOT-NULL flag per selected row, both counts.
    Pages of every awkward size incl. empty and all-NULL ones, page starts inside selection words, code widths
    that grow along the chunk (several runs), FLE values and dictionary entries of 4 and 8 bytes; against the
    row model."""
    rng = np.random.default_rng(1200 + seed)
    n = int(rng.choice([1, 77, 5000, 70001, 400009]))
    sizes = [RAGGED, [2048, 4096], [1, 5, 31, 33, 64], [70001, 10000, 0], [65536 * 4 + 3, 100], RAGGED][seed]
    page_rows = cuts(rng, n, sizes)
    null_frac = [0.2, 0.0, 0.5, 0.9, 0.1, 1.0][seed]
    is_set = rng.random(n) >= null_frac
    use_dict = seed % 3  # 0: FLE values, 1: int32 dictionary, 2: int64 dictionary
    if use_dict:
        t = capi.T_INT32 if use_dict == 1 else capi.T_INT64
        pool = np.sort(rng.choice(np.arange(-10 ** 6, 10 ** 6), 300, replace=False)).astype(O.NP_TYPES[t])
        codes = rng.integers(0, len(pool), n).astype(np.uint32)
        row_vals = pool[codes]
        d = capi.Dict(O.plain_encode(pool, t), t)
        widths = [9, 9, 10]  # (the writer's width grows with the dictionary; the codes fit all of them)
    else:
        codes = rng.integers(0, 1 << 7, n).astype(np.uint32)
        row_vals = codes
        d = None
        widths = [7, 7, 11]
    pages, pos = [], 0
    for i, m in enumerate(page_rows):
        w = widths[min(i * 3 // max(len(page_rows), 1), 2)]
        if m == 0:
            pages.append((None, 0, w, torch.zeros(2, dtype=torch.int64, device="cuda"), 0))
            continue
        s = is_set[pos:pos + m]
        k = int(s.sum())
        defs = O.fle_encode(s.astype(np.uint32), 1)
        enc = O.fle_encode(codes[pos:pos + m][s], w) if k else np.zeros(2, np.uint64)
        pages.append((dev_words(enc), m, w, dev_words(defs), k))
        pos += m
    ch = capi.Chunk(pages, max_def_level=1)
    for density in (0.1, 1.0, 0.0, 0.6):
        sel = rng.random(n) < density
        if density == 1.0:
            sel[:] = True
        bm = torch.from_numpy(np.concatenate([pack(sel), np.zeros(2, np.uint64)]).view(np.int64)).cuda()
        dense, flags, n_sel, n_val, bad = ch.select_nullable(bm, d)
        assert bad == 0
        assert n_sel == int(sel.sum()) and n_val == int((sel & is_set).sum()), (seed, density, n)
        exp = row_vals[sel & is_set]
        got = dense.cpu().numpy()
        if not use_dict:
            got = got.view(np.uint32)
        assert np.array_equal(got, exp), (seed, density, n, page_rows[:6])
        assert np.array_equal(bits_of(words(flags), n_sel), is_set[sel]), (seed, density, n)
    ch.close()
    if d:
        d.close()

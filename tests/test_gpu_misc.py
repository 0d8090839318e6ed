"""GPU parity tests of the dictionary path, PLAIN-page predicates, bitmap algebra, the
IntersectBitset expand and the fused predicate program, all through the C-ABI vs the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev_words(a):
    a = np.ascontiguousarray(a)
    if a.size == 0:
        return torch.zeros(2, dtype=torch.int64, device="cuda")
    return torch.from_numpy(a.view(np.int64)).cuda()


def words(t):
    return t.cpu().numpy().view(np.uint64)


def bits_of(w, n):
    return np.unpackbits(np.ascontiguousarray(w).view(np.uint8), bitorder="little")[:n].astype(bool)


def page_blocks(data_page):
    """DictDecoderBase::SetData strips the 1-byte width header (dict-encoding.h:185-192)."""
    return int(data_page[0]), np.frombuffer(data_page[1:].tobytes(), dtype=np.uint64)


TYPES = ["T_INT8", "T_INT16", "T_INT32", "T_INT64", "T_FLOAT", "T_DOUBLE"]


def make_dict_column(O, t, rng, n, distinct):
    npt = O.NP_TYPES[t]
    if np.issubdtype(npt, np.integer):
        info = np.iinfo(npt)
        lo, hi = max(info.min, -(10 ** 9)), min(info.max, 10 ** 9)
        pool = rng.choice(np.arange(lo, hi, max((hi - lo) // (4 * distinct), 1)), distinct, replace=False)
    else:
        pool = np.unique(rng.normal(0, 1000, 2 * distinct).astype(npt))[:distinct]
    pool = pool.astype(npt)
    vals = pool[rng.integers(0, len(pool), n)]
    vals[:len(pool)] = pool
    return vals


@pytest.mark.parametrize("type_name", TYPES)
def test_dict_pred_decode_scan(capi, O, type_name):
    t = getattr(O, type_name)
    rng = np.random.default_rng(40 + t)
    distinct = 100 if t == O.T_INT8 else 300
    n = 5000
    vals = make_dict_column(O, t, rng, n, distinct)
    d, dict_page, codes = O.dict_build(vals, t)
    data = O.dict_write_data(codes, len(d))
    bw, blocks = page_blocks(data)
    assert bw == capi.dict_bit_width(len(d))
    dd = capi.Dict(dict_page, t)
    assert dd.num_entries == len(d)
    enc = dev_words(blocks)

    out, bad = dd.decode(enc, n, bw)
    got = out.cpu().numpy().astype(O.NP_TYPES[t])
    assert int(bad.item()) == 0 and np.array_equal(got, vals)

    below = d[0] - 1 if d[0] > np.finfo(np.float32).min else d[0]
    absent = (d[10] + d[11]) / 2 if not np.issubdtype(d.dtype, np.integer) else None
    lits = [below, d[0], d[len(d) // 2], d[-1], d[-1] + 1]
    if absent is not None:
        lits.append(absent)
    lits = np.array(lits).astype(O.NP_TYPES[t])
    for lit in lits:
        for op in (O.OP_EQ, O.OP_LT, O.OP_LE, O.OP_GT, O.OP_GE):
            kind, fle_op, xcodes = O.dict_translate(d, t, op, lit)
            got_kind, got_op, got_codes = dd.translate(op, lit)
            assert got_kind == kind
            if kind == O.XL_FLE:
                assert (got_op, got_codes) == (fle_op, [int(x) for x in xcodes])
            ref = O.dict_pred(d, t, data, n, op, lit)
            assert np.array_equal(words(dd.pred(enc, n, bw, op, lit)), ref), (type_name, op, lit)
            bitmap, bvals, counts = dd.scan(enc, n, bw, op, lit)
            assert np.array_equal(words(bitmap), ref)
            dense = capi.batches_compact(bvals, counts, n).cpu().numpy().astype(O.NP_TYPES[t])
            assert np.array_equal(dense, vals[bits_of(ref, n)]), (type_name, op, lit)
    in_list = np.array([d[3], d[-2], d[0] - 1 if np.issubdtype(d.dtype, np.integer) else d[0] - 0.5,
                        d[7]]).astype(O.NP_TYPES[t])
    ref = O.dict_pred(d, t, data, n, O.OP_IN, in_list)
    assert np.array_equal(words(dd.pred(enc, n, bw, O.OP_IN, in_list)), ref)
    bitmap, bvals, counts = dd.scan(enc, n, bw, O.OP_IN, in_list)
    assert np.array_equal(words(bitmap), ref)
    dense = capi.batches_compact(bvals, counts, n).cpu().numpy().astype(O.NP_TYPES[t])
    assert np.array_equal(dense, vals[bits_of(ref, n)])
    dd.close()


@pytest.mark.parametrize("type_name", ["T_INT32", "T_INT64", "T_DOUBLE"])
@pytest.mark.parametrize("distinct", [5, 60, 256])
def test_narrow_dictionary_scan_every_selectivity(capi, O, type_name, distinct):
    """Dictionaries of up to 256 entries (code widths 3 / 6 / 8: the scan's 4 KiB LDS layout) with
    4- and 8-byte entries: fused scan + gather and select from 0 % to 100 % selectivity, i.e. the
    gather path, the index list and the packed dense path, over ragged multi-batch columns."""
    t = getattr(O, type_name)
    rng = np.random.default_rng(900 + t + distinct)
    for n in (257, 2047, 2049, 20011):
        vals = make_dict_column(O, t, rng, n, distinct)
        d, dict_page, codes = O.dict_build(vals, t)
        data = O.dict_write_data(codes, len(d))
        bw, blocks = page_blocks(data)
        dd = capi.Dict(dict_page, t)
        enc = dev_words(blocks)
        out, bad = dd.decode(enc, n, bw)   # the packed decode path: 4- and 8-byte entries, ragged n
        assert int(bad.item()) == 0 and np.array_equal(out.cpu().numpy().astype(O.NP_TYPES[t]), vals)
        for q in (0.0, 0.02, 0.2, 0.3, 0.6, 1.0):
            lit = d[min(len(d) - 1, int(q * len(d)))]
            for op in ((O.OP_LT, O.OP_GE) if q < 1.0 else (O.OP_LE,)):
                ref = O.dict_pred(d, t, data, n, op, lit)
                bitmap, bvals, counts = dd.scan(enc, n, bw, op, lit)
                assert np.array_equal(words(bitmap), ref), (type_name, distinct, n, q, op)
                dense = capi.batches_compact(bvals, counts, n).cpu().numpy().astype(O.NP_TYPES[t])
                assert np.array_equal(dense, vals[bits_of(ref, n)]), (type_name, distinct, n, q, op)
                sv, sc = dd.select(enc, n, bw, bitmap)
                assert torch.equal(sc, counts)
                dense2 = capi.batches_compact(sv, sc, n).cpu().numpy().astype(O.NP_TYPES[t])
                assert np.array_equal(dense2, dense)
        dd.close()


def test_dict_bad_index_flag(capi, O):
    """A code >= num_entries: the reference's GetValue returns false (dict-encoding.h:316)."""
    d = np.arange(5, dtype=np.int32)
    dd = capi.Dict(d.view(np.uint8), O.T_INT32)
    codes = np.array([0, 1, 7, 2] * 20, dtype=np.uint32)      # 7 >= 5
    enc = dev_words(O.fle_encode(codes, 3))
    _, bad = dd.decode(enc, len(codes), 3)
    assert int(bad.item()) != 0
    dd.close()


@pytest.mark.parametrize("type_name", TYPES)
def test_plain_pred(capi, O, type_name):
    t = getattr(O, type_name)
    rng = np.random.default_rng(60 + t)
    npt = O.NP_TYPES[t]
    for n in (1, 63, 64, 65, 255, 256, 257, 5000, 20011):
        if np.issubdtype(npt, np.integer):
            info = np.iinfo(npt)
            vals = rng.integers(max(info.min, -500), min(info.max, 500), n).astype(npt)
        else:
            vals = rng.normal(0, 100, n).astype(npt)
        page = O.plain_encode(vals, t)
        d_page = torch.from_numpy(np.concatenate([page, np.zeros(16, np.uint8)])).cuda()
        lit = vals[n // 2]
        for op in (O.OP_EQ, O.OP_LT, O.OP_LE, O.OP_GT, O.OP_GE):
            for sem in (O.SEM_REFERENCE, O.SEM_SQL):
                got = words(capi.plain_pred(d_page, n, t, op, lit, sem))
                assert np.array_equal(got, O.plain_pred(page, n, t, op, lit, sem)), (type_name, n, op, sem)
        lst = np.array([vals[0], vals[-1], 499]).astype(npt)
        got = words(capi.plain_pred(d_page, n, t, O.OP_IN, lst, O.SEM_SQL))
        assert np.array_equal(got, O.plain_pred(page, n, t, O.OP_IN, lst, O.SEM_SQL))
    with pytest.raises(capi.IpsError):      # empty reference body, parquet-common.h:252-255
        capi.plain_pred(d_page, n, t, O.OP_IN, lst, O.SEM_REFERENCE)


def test_bitmap_algebra_and_expand(capi, O):
    rng = np.random.default_rng(7)
    for n in (1, 64, 65, 1000, 70001):
        nw = (n + 63) // 64
        a = rng.integers(0, 2 ** 63, nw).astype(np.uint64) * 2 + rng.integers(0, 2, nw).astype(np.uint64)
        b = rng.integers(0, 2 ** 63, nw).astype(np.uint64) * 2 + rng.integers(0, 2, nw).astype(np.uint64)
        tail = n - (nw - 1) * 64
        if tail < 64:
            a[-1] &= np.uint64((1 << tail) - 1)
            b[-1] &= np.uint64((1 << tail) - 1)
        assert np.array_equal(words(capi.bitmap_and(dev_words(a), dev_words(b), n)), a & b)
        assert np.array_equal(words(capi.bitmap_or(dev_words(a), dev_words(b), n)), a | b)
        assert capi.bitmap_count(dev_words(a), n) == int(bits_of(a, n).sum())
        ones = words(capi.bitmap_fill(dev_words(a), n, 1))
        assert bits_of(ones, n).all() and (tail == 64 or int(ones[-1]) >> tail == 0)
        assert not words(capi.bitmap_fill(dev_words(a), n, 0)).any()
        # IntersectBitset (hdfs-parquet-scanner.cc:326-331)
        k = int(bits_of(a, n).sum())
        sub_bits = rng.random(max(k, 1)) < 0.4
        sub = np.packbits(sub_bits, bitorder="little")
        sub = np.concatenate([sub, np.zeros(-len(sub) % 8, np.uint8)]).view(np.uint64)
        got = words(capi.bitmap_expand(dev_words(a), dev_words(sub), n))
        assert np.array_equal(got, O.bitmap_expand(a, sub, n)), n


def test_nullable_dictionary_column(capi, O):
    """ColumnReader::Lt on an OPTIONAL dictionary column (hdfs-parquet-scanner.cc:352-369):
    nonnull = def_levels.Eq(max_def); data = dict.Lt over count(nonnull) values; IntersectBitset."""
    rng = np.random.default_rng(17)
    n = 30011
    is_set = rng.random(n) < 0.8
    data_vals = rng.integers(-1000, 1000, int(is_set.sum())).astype(np.int32)
    defs = O.fle_encode(is_set.astype(np.uint32), 1)        # FLE def levels, bw = Log2(1+1) = 1
    d, dict_page, codes = O.dict_build(data_vals, O.T_INT32)
    bw, blocks = page_blocks(O.dict_write_data(codes, len(d)))
    dd = capi.Dict(dict_page, O.T_INT32)
    nonnull = capi.fle_pred(dev_words(defs), n, 1, O.OP_EQ, 1)
    k = capi.bitmap_count(nonnull, n)
    assert k == len(data_vals)
    data_bm = dd.pred(dev_words(blocks), k, bw, O.OP_LT, np.int32(-200))
    got = bits_of(words(capi.bitmap_expand(nonnull, data_bm, n)), n)
    expect = np.zeros(n, bool)
    expect[np.flatnonzero(is_set)[data_vals < -200]] = True
    assert np.array_equal(got, expect)
    dd.close()


@pytest.mark.parametrize("strategy", ["auto", "general", "one_pass"])
def test_fused_program(capi, O, strategy, request):
    """EvalSimplePredicates over several columns vs numpy: BETWEEN = And(Ge, Le); And(Gt a, Lt b);
    Or; IN; PLAIN leaves (int32, int64, double).  'auto' lets ips_eval_program use the
    per-operand plan (stand-alone predicate kernels on a stack of bitmaps); 'general' forces the one-launch
    program kernel for every tree."""
    # 'one_pass': chains of <= 4 operands run as the one-pass kernel, the rest as planned
    request.addfinalizer(lambda: capi.set_program_strategy(capi.PROGRAM_AUTO))
    capi.set_program_strategy({"auto": capi.PROGRAM_AUTO, "general": capi.PROGRAM_ONE_LAUNCH,
                               "one_pass": capi.PROGRAM_ONE_PASS}[strategy])
    rng = np.random.default_rng(23)
    for n in (1, 2047, 2049, 50021):
        c0 = rng.integers(0, 1 << 12, n).astype(np.uint32)
        c1 = rng.integers(0, 1 << 4, n).astype(np.uint32)
        c2 = rng.integers(0, 1 << 21, n).astype(np.uint32)
        p32 = rng.integers(-1000, 1000, n).astype(np.int32)
        p64 = rng.integers(-10 ** 12, 10 ** 12, n).astype(np.int64)
        pf64 = rng.normal(0, 10, n)
        e0, e1, e2 = O.fle_encode(c0, 12), O.fle_encode(c1, 4), O.fle_encode(c2, 21)
        pad = np.zeros(16, np.uint8)
        d32 = torch.from_numpy(np.concatenate([p32.view(np.uint8), pad])).cuda()
        d64 = torch.from_numpy(np.concatenate([p64.view(np.uint8), pad])).cuda()
        df64 = torch.from_numpy(np.concatenate([pf64.view(np.uint8), pad])).cuda()
        de0, de1, de2 = dev_words(e0), dev_words(e1), dev_words(e2)
        cols = [capi.fle_column(de0, 12), capi.fle_column(de1, 4), capi.fle_column(de2, 21),
                capi.plain_column(d32, O.T_INT32), capi.plain_column(d64, O.T_INT64),
                capi.plain_column(df64, O.T_DOUBLE)]
        L, PL, AND, OR = capi.leaf, capi.plain_leaf, capi.and_node, capi.or_node
        # Q6-shaped: c0 >= 1000 AND c0 < 3000 AND (c1 BETWEEN 5 AND 7) AND c2 < 2^20
        nodes = [L(0, O.OP_GE, 1000), L(0, O.OP_LT, 3000), AND(),
                 L(1, O.OP_GE, 5), L(1, O.OP_LE, 7), AND(), AND(),
                 L(2, O.OP_LT, 1 << 20), AND()]
        got = bits_of(words(capi.eval_program(nodes, cols, n)), n)
        exp = (c0 >= 1000) & (c0 < 3000) & (c1 >= 5) & (c1 <= 7) & (c2 < (1 << 20))
        assert np.array_equal(got, exp), n
        # mixed encodings with OR and IN
        nodes = [PL(3, O.OP_GT, np.int32(100), O.T_INT32), PL(4, O.OP_LE, np.int64(0), O.T_INT64), OR(),
                 L(1, O.OP_IN, [1, 9, 15]), AND(),
                 PL(5, O.OP_LT, np.float64(-3.5), O.T_DOUBLE), OR()]
        got = bits_of(words(capi.eval_program(nodes, cols, n)), n)
        exp = (((p32 > 100) | (p64 <= 0)) & np.isin(c1, [1, 9, 15])) | (pf64 < -3.5)
        assert np.array_equal(got, exp), n
        # (A and B) or (C and D): two live bitmaps
        nodes = [L(0, O.OP_LT, 2000), PL(3, O.OP_GE, np.int32(0), O.T_INT32), AND(),
                 L(2, O.OP_GE, 1 << 19), PL(4, O.OP_LT, np.int64(5), O.T_INT64), AND(), OR()]
        got = bits_of(words(capi.eval_program(nodes, cols, n)), n)
        exp = ((c0 < 2000) & (p32 >= 0)) | ((c2 >= (1 << 19)) & (p64 < 5))
        assert np.array_equal(got, exp), n
        # leaf OP (subtree): the root ends up in the later bitmap; three bitmaps live at once
        nodes = [L(1, O.OP_GT, 2),
                 L(0, O.OP_LT, 900), PL(3, O.OP_LT, np.int32(-500), O.T_INT32), OR(),
                 L(2, O.OP_GE, 1 << 20), PL(4, O.OP_GT, np.int64(0), O.T_INT64), AND(),
                 PL(5, O.OP_GT, np.float64(12.0), O.T_DOUBLE), L(0, O.OP_GE, 4000), OR(),
                 OR(), AND(), AND()]
        got = bits_of(words(capi.eval_program(nodes, cols, n)), n)
        exp = (c1 > 2) & (((c0 < 900) | (p32 < -500)) &
                          (((c2 >= (1 << 20)) & (p64 > 0)) | ((pf64 > 12.0) | (c0 >= 4000))))
        assert np.array_equal(got, exp), n
        # BETWEEN on a PLAIN column + OR-chain
        nodes = [PL(5, O.OP_GE, np.float64(-1.0), O.T_DOUBLE), PL(5, O.OP_LE, np.float64(2.5), O.T_DOUBLE),
                 AND(), L(1, O.OP_EQ, 3), OR(), PL(3, O.OP_EQ, np.int32(p32[0]), O.T_INT32), OR()]
        got = bits_of(words(capi.eval_program(nodes, cols, n)), n)
        exp = ((pf64 >= -1.0) & (pf64 <= 2.5)) | (c1 == 3) | (p32 == p32[0])
        assert np.array_equal(got, exp), n
        # a single leaf equals ips_fle_pred
        nodes = [L(2, O.OP_EQ, int(c2[0]))]
        assert np.array_equal(words(capi.eval_program(nodes, cols, n)),
                              O.fle_pred(e2, n, 21, O.OP_EQ, int(c2[0])))
    with pytest.raises(capi.IpsError):
        capi.eval_program([capi.and_node()], cols, n)            # stack underflow is an error


def test_assemble_tuples(capi, O):
    """Multi-column late materialisation (AssembleRows' vector path, scanner.cc:1151-1181): three
    columns selected by one conjunction, written as row-major tuples in row order."""
    rng = np.random.default_rng(31)
    n = 30011
    a = rng.integers(0, 1 << 12, n).astype(np.uint32)             # FLE int32-like
    d64 = np.sort(rng.choice(np.arange(-10 ** 12, 10 ** 12, 10 ** 7), 500, replace=False)).astype(np.int64)
    codes = rng.integers(0, 500, n).astype(np.uint32)             # dictionary int64
    c = rng.integers(0, 1 << 20, n).astype(np.uint32)
    ea, ec = dev_words(O.fle_encode(a, 12)), dev_words(O.fle_encode(c, 20))
    bw = capi.dict_bit_width(500)
    ecodes = dev_words(O.fle_encode(codes, bw))
    dd = capi.Dict(d64.view(np.uint8), O.T_INT64)
    # predicate on a and c -> one bitmap; then every column is materialised against it
    nodes = [capi.leaf(0, O.OP_LT, 1000), capi.leaf(1, O.OP_GE, 1 << 19), capi.and_node()]
    bm = capi.eval_program(nodes, [capi.fle_column(ea, 12), capi.fle_column(ec, 20)], n)
    sel = (a < 1000) & (c >= (1 << 19))
    va, counts = capi.fle_select(ea, n, 12, bm)
    vc, counts2 = capi.fle_select(ec, n, 20, bm)
    assert torch.equal(counts, counts2)
    v64, counts3 = dd.select(ecodes, n, bw, bm)      # dict[code] of the selected rows only
    assert torch.equal(counts, counts3)
    tuple_size = 24   # [int32 a @0][pad][int64 d @8][int32 c @16][pad]
    tuples = capi.assemble_tuples([(va, 0), (v64, 8), (vc, 16)], counts, n, tuple_size)
    t = tuples.cpu().numpy()
    assert t.shape == (int(sel.sum()), tuple_size)
    assert np.array_equal(t[:, 0:4].copy().view(np.uint32).ravel(), a[sel])
    assert np.array_equal(t[:, 8:16].copy().view(np.int64).ravel(), d64[codes][sel])
    assert np.array_equal(t[:, 16:20].copy().view(np.uint32).ravel(), c[sel])
    assert not t[:, 4:8].any() and not t[:, 20:24].any()
    # InitTuple(): every tuple starts as a copy of the template tuple; slots overwrite it
    tmpl = rng.integers(1, 255, tuple_size).astype(np.uint8)
    t = capi.assemble_tuples([(va, 0), (v64, 8), (vc, 16)], counts, n, tuple_size, template=tmpl).cpu().numpy()
    assert np.array_equal(t[:, 0:4].copy().view(np.uint32).ravel(), a[sel])
    assert np.array_equal(t[:, 16:20].copy().view(np.uint32).ravel(), c[sel])
    assert (t[:, 4:8] == tmpl[4:8]).all() and (t[:, 20:24] == tmpl[20:24]).all()
    # tuple layouts off the staged path: odd size, unaligned slots (direct writes), wide tuples
    for size, offs in ((23, (1, 11, 19)), (200, (4, 104, 196)), (132, (0, 64, 128))):
        tmpl = rng.integers(1, 255, size).astype(np.uint8)
        t = capi.assemble_tuples([(va, offs[0]), (v64, offs[1]), (vc, offs[2])], counts, n, size,
                                 template=tmpl).cpu().numpy()
        assert t.shape == (int(sel.sum()), size)
        assert np.array_equal(t[:, offs[0]:offs[0] + 4].copy().view(np.uint32).ravel(), a[sel])
        assert np.array_equal(t[:, offs[1]:offs[1] + 8].copy().view(np.int64).ravel(), d64[codes][sel])
        assert np.array_equal(t[:, offs[2]:offs[2] + 4].copy().view(np.uint32).ravel(), c[sel])
        covered = np.zeros(size, bool)
        for o, w in ((offs[0], 4), (offs[1], 8), (offs[2], 4)):
            covered[o:o + w] = True
        assert (t[:, ~covered] == tmpl[~covered]).all()
    dd.close()


def test_native_rccl_allgather_single_rank(capi, O):
    """ips_allgather_bitmap through librccl (dlopen'ed on first use) with a 1-rank communicator:
    the gathered words equal the local words.  (N > 1 is exercised by bench.py --gpus N and, for
    the host logic, by tests/test_distributed_cpu.py.)"""
    vals = np.arange(10000, dtype=np.uint32) % 97
    enc = dev_words(O.fle_encode(vals, 7))
    bm = capi.fle_pred(enc, len(vals), 7, O.OP_LT, 13)
    comm = capi.Comm(capi.comm_unique_id(), 1, 0)
    full = comm.allgather_bitmap(bm)
    torch.cuda.synchronize()
    assert torch.equal(full, bm)
    comm.close()


def test_bitmap_compress(capi, O):
    rng = np.random.default_rng(41)
    for n in (1, 64, 65, 1000, 70001):
        nw = (n + 63) // 64
        mask_bits = rng.random(n) < 0.6
        src_bits = rng.random(n) < 0.3
        pack = lambda b: np.concatenate([np.packbits(b, bitorder="little"),
                                         np.zeros(-((len(b) + 7) // 8) % 8, np.uint8)]).view(np.uint64)
        out, k = capi.bitmap_compress(dev_words(pack(mask_bits)), dev_words(pack(src_bits)), n)
        assert k == int(mask_bits.sum())
        got = bits_of(words(out), n)
        exp = np.zeros(n, bool)
        exp[:k] = src_bits[mask_bits]
        assert np.array_equal(got, exp), n
        # compress is the inverse of expand on the mask's support
        back = capi.bitmap_expand(dev_words(pack(mask_bits)), out, n)
        assert np.array_equal(bits_of(words(back), n), src_bits & mask_bits)
        assert words(out).shape[0] == nw


def test_nullable_column_materialisation(capi, O):
    """OPTIONAL dictionary column next to a REQUIRED one: the predicate selects rows, some of
    which are NULL in the OPTIONAL column; tuples carry the value or the NULL indicator bit
    (ReadValue: ReadDefinitionLevel + ReadSlot(skip), hdfs-parquet-scanner.cc:1006-1027)."""
    rng = np.random.default_rng(43)
    n = 40013
    a = rng.integers(0, 1 << 10, n).astype(np.uint32)              # REQUIRED FLE column
    is_set = rng.random(n) < 0.75
    data_vals = rng.integers(-500, 500, int(is_set.sum())).astype(np.int32)
    d, dict_page, codes = O.dict_build(data_vals, O.T_INT32)
    bw, blocks = page_blocks(O.dict_write_data(codes, len(d)))
    dd = capi.Dict(dict_page, O.T_INT32)
    ea = dev_words(O.fle_encode(a, 10))
    defs = dev_words(O.fle_encode(is_set.astype(np.uint32), 1))
    n_data = len(data_vals)

    sel = capi.fle_pred(ea, n, 10, O.OP_LT, 300)                  # selection from the REQUIRED column
    nonnull = capi.fle_pred(defs, n, 1, O.OP_EQ, 1)               # def level == max_def_level
    data_sel, k = capi.bitmap_compress(nonnull, sel, n)           # selection over the data rows
    assert k == n_data
    vb, cb = dd.select(dev_words(blocks), n_data, bw, data_sel)
    dense = capi.batches_compact(vb, cb, n_data)                  # values of selected non-NULL rows
    flags, n_sel = capi.bitmap_compress(sel, nonnull, n)          # non-NULL flag per selected row
    va, counts = capi.fle_select(ea, n, 10, sel)
    tuple_size = 12                                               # [u32 a @0][i32 opt @4][null byte @8]
    tuples = capi.assemble_tuples([(va, 0), (dense, 4, flags, 8, 0x01)], counts, n, tuple_size)
    t = tuples.cpu().numpy()

    sel_np = a < 300
    assert t.shape[0] == int(sel_np.sum()) == n_sel
    full = np.zeros(n, np.int32)
    full[is_set] = data_vals
    assert np.array_equal(t[:, 0:4].copy().view(np.uint32).ravel(), a[sel_np])
    nulls = ~is_set[sel_np]
    assert np.array_equal(t[:, 8] & 1, nulls.astype(np.uint8))
    got_opt = t[:, 4:8].copy().view(np.int32).ravel()
    assert np.array_equal(got_opt[~nulls], full[sel_np][~nulls])
    assert not got_opt[nulls].any()                               # NULL slots untouched (zero)
    # the same in ONE call per OPTIONAL column (ips_dict_select_nullable), also on the level words
    n_data_blocks = ((n_data + 63) // 64) * 64
    dense1, flags1, n_sel1, n_val1 = capi.select_nullable(dd, defs, 1, 1, n, dev_words(blocks), n_data_blocks, bw, sel)
    assert n_sel1 == n_sel and n_val1 == dense.numel()
    assert torch.equal(dense1, dense)
    assert torch.equal(flags1[:(n_sel + 63) // 64], flags[:(n_sel + 63) // 64])
    tuples1 = capi.assemble_tuples([(va, 0), (dense1, 4, flags1, 8, 0x01)], counts, n, tuple_size)
    assert torch.equal(tuples1, tuples)
    dd.close()


@pytest.mark.parametrize("type_name", TYPES)
def test_dict_encode_on_gpu(capi, O, type_name):
    """GPU DictEncoder (Put / WriteDict / WriteData) vs the oracle's sort + remap, and the
    round trip through the GPU DictDecoder (dict-test.cc's ValidateDict shape)."""
    t = getattr(O, type_name)
    rng = np.random.default_rng(70 + t)
    for n, distinct in ((1, 1), (100, 1), (5000, 300), (200000, 39999 if t not in (O.T_INT8,) else 200)):
        distinct = min(distinct, 100) if t == O.T_INT8 else distinct
        vals = make_dict_column(O, t, rng, max(n, distinct), distinct)[:max(n, distinct)]
        n = len(vals)
        d_ref, page_ref, codes_ref = O.dict_build(vals, t)
        slots = O.plain_encode(vals, t)                      # 4- or 8-byte PLAIN slots
        if t in (O.T_INT8, O.T_INT16):
            # the reference's writer leaves the upper slot bytes unspecified (parquet-common.h:170)
            junk = rng.integers(0, 256, len(slots), dtype=np.uint8)
            keep = np.arange(len(slots)) % 4 < O.NP_TYPES[t]().itemsize
            slots = np.where(keep, slots, junk).astype(np.uint8)
        d_slots = torch.from_numpy(np.concatenate([slots, np.zeros(16, np.uint8)])).cuda()
        page, bw, enc = capi.dict_encode(d_slots[:len(slots)].view(
            {4: torch.int32, 8: torch.int64}[len(slots) // n]), t)
        assert np.array_equal(page, page_ref), (type_name, n)
        assert bw == O.bit_width_for_entries(len(d_ref))
        if bw:
            assert np.array_equal(words(enc), O.fle_encode(codes_ref, bw))
            dd = capi.Dict(page, t)
            out, bad = dd.decode(enc, n, bw)
            assert int(bad.item()) == 0
            assert np.array_equal(out.cpu().numpy().astype(O.NP_TYPES[t]), vals)
            dd.close()
    if t == O.T_INT32:
        page, bw, enc = capi.dict_encode(torch.zeros(16, dtype=torch.int32, device="cuda")[:0], t)
        assert len(page) == 0 and bw == 0 and enc.numel() == 0      # empty chunk
    # the 40000-entry cap of the reference
    if t == O.T_INT32:
        big = torch.arange(0, 50000, dtype=torch.int32, device="cuda")
        with pytest.raises(capi.IpsError):
            capi.dict_encode(big, t)


@pytest.mark.parametrize("strategy", ["auto", "general", "one_pass"])
def test_random_predicate_trees(capi, O, strategy, request):
    """Random AND/OR trees (up to 12 leaves, any shape) over FLE and PLAIN columns against numpy:
    exercises the per-operand planner's bitmap stack (which bitmap ends up as the root, temporaries,
    same-column pairs, IN leaves) and the one-launch interpreter on the same programs."""
    # 'one_pass': chains of <= 4 operands run as the one-pass kernel, the rest as planned
    request.addfinalizer(lambda: capi.set_program_strategy(capi.PROGRAM_AUTO))
    capi.set_program_strategy({"auto": capi.PROGRAM_AUTO, "general": capi.PROGRAM_ONE_LAUNCH,
                               "one_pass": capi.PROGRAM_ONE_PASS}[strategy])
    rng = np.random.default_rng(4242)
    n = 20000 + 37
    widths = (5, 12, 32)   # 32: the early-pruning kernel, also in its and-into / or-into modes
    fle_np = [rng.integers(0, 1 << w, n, dtype=np.uint64).astype(np.uint32) for w in widths]
    fle_np[2][::3] &= 0xFFFF   # rows that stay undecided after the high planes
    p32 = rng.integers(-100, 100, n).astype(np.int32)
    p64 = rng.integers(-10 ** 6, 10 ** 6, n).astype(np.int64)
    encs = [dev_words(O.fle_encode(v, w)) for v, w in zip(fle_np, widths)]
    pad = np.zeros(16, np.uint8)
    d32 = torch.from_numpy(np.concatenate([p32.view(np.uint8), pad])).cuda()
    d64 = torch.from_numpy(np.concatenate([p64.view(np.uint8), pad])).cuda()
    cols = [capi.fle_column(e, w) for e, w in zip(encs, widths)] + \
           [capi.plain_column(d32, O.T_INT32), capi.plain_column(d64, O.T_INT64)]
    data = fle_np + [p32, p64]
    ops = {O.OP_EQ: np.equal, O.OP_LT: np.less, O.OP_LE: np.less_equal, O.OP_GT: np.greater,
           O.OP_GE: np.greater_equal}

    def random_leaf():
        col = int(rng.integers(0, 5))
        x = data[col]
        if col < 3 and rng.random() < 0.25:
            lst = [int(v) for v in rng.choice(x, int(rng.integers(1, 13)))]
            return capi.leaf(col, O.OP_IN, lst), np.isin(x, lst)
        op = int(rng.choice(list(ops)))
        lit = x[int(rng.integers(0, n))]
        if col < 3:
            return capi.leaf(col, op, int(lit)), ops[op](x, lit)
        t = O.T_INT32 if col == 3 else O.T_INT64
        return capi.plain_leaf(col, op, lit, t), ops[op](x, lit)

    def random_tree(leaves):
        """-> (postfix nodes, numpy truth, stack depth needed)"""
        if leaves == 1:
            nd, truth = random_leaf()
            return [nd], truth, 1
        left = int(rng.integers(1, leaves))
        a, ta, da = random_tree(left)
        b, tb, db = random_tree(leaves - left)
        if rng.random() < 0.5:
            return a + b + [capi.and_node()], ta & tb, max(da, db + 1)
        return a + b + [capi.or_node()], ta | tb, max(da, db + 1)

    done = 0
    while done < 25:
        nodes, truth, depth = random_tree(int(rng.integers(1, 13)))
        if depth > 8:            # the interpreter's stack (both strategies accept <= 8)
            continue
        got = bits_of(words(capi.eval_program(nodes, cols, n)), n)
        assert np.array_equal(got, truth), (strategy, done, len(nodes))
        done += 1


def test_entry_points_capture_into_a_graph(capi, O):
    """The launch-only entry points neither allocate nor synchronise, so a host can capture a
    page's whole evaluation into a hipGraph and replay it (bench.py does so for the 2^20-row
    latency figure): predicate, fused scan, the page-list scan and ips_eval_program -- including
    a tree that needs a temporary bitmap -- captured once, replayed on new data."""
    rng = np.random.default_rng(77)
    n = 50000
    a = rng.integers(0, 1 << 12, n).astype(np.uint32)
    b = rng.integers(0, 1 << 6, n).astype(np.uint32)
    ea, eb = dev_words(O.fle_encode(a, 12)), dev_words(O.fle_encode(b, 6))
    outs = capi.alloc_scan_outputs(n, torch.device("cuda"))
    bm_pred = torch.zeros((n + 63) // 64 + 2, dtype=torch.int64, device="cuda")
    bm_chain = torch.zeros_like(bm_pred)
    bm_tree = torch.zeros_like(bm_pred)
    cols = [capi.fle_column(ea, 12), capi.fle_column(eb, 6)]
    L, AND, OR = capi.leaf, capi.and_node, capi.or_node
    chain = [L(0, O.OP_GE, 100), L(0, O.OP_LT, 3000), AND(), L(1, O.OP_LT, 40), AND()]
    tree = [L(0, O.OP_LT, 500), L(1, O.OP_GE, 10), AND(), L(0, O.OP_GE, 3500), L(1, O.OP_LT, 5), AND(), OR()]
    pages = [(ea, n, capi.alloc_scan_outputs(n, torch.device("cuda")))]
    plist = capi.make_page_list(pages)
    # the tree keeps two bitmaps alive: its temporary lives in a caller workspace, so the captured
    # launches are the per-operand plan (no hidden allocation, nothing library-owned baked in)
    assert capi.program_workspace_bytes(chain, cols, n) == 0
    need = capi.program_workspace_bytes(tree, cols, n)
    assert need >= ((n + 63) // 64) * 8
    ws_tree = torch.empty(need, dtype=torch.uint8, device="cuda")
    with pytest.raises(capi.IpsError):      # a tree that needs a workspace does not get one
        import ctypes as C
        arr_n = (capi.Node * len(tree))(*tree)
        arr_c = (capi.Column * len(cols))(*cols)
        capi._ck(capi.lib().ips_eval_program(arr_n, len(tree), arr_c, len(cols), C.c_int64(n),
                                             C.c_void_p(bm_tree.data_ptr()), None, capi._stream()))

    # the same chain over the two columns as page lists (one page each, the very buffers above): the one-pass
    # chain with blockIdx.y = page
    chunks = [capi.Chunk([(ea, n, 12)]), capi.Chunk([(eb, n, 6)])]
    bm_pages = torch.zeros_like(bm_pred)

    def work():
        capi.fle_pred(ea, n, 12, O.OP_LT, 1000, bitmap=bm_pred)
        capi.fle_scan(ea, n, 12, O.OP_LT, 1000, outputs=outs)
        capi.fle_scan_pages(plist, 12, O.OP_GE, 2000)
        capi.eval_program(chain, cols, n, bitmap=bm_chain)
        capi.eval_program(tree, cols, n, bitmap=bm_tree, workspace=ws_tree)
        capi.eval_program_chunks(chain, chunks, bitmap=bm_pages)

    work()                                   # warm-up outside the capture (module loading etc.)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            work()
    torch.cuda.current_stream().wait_stream(side)
    # new data in the same buffers, then replay
    a2 = rng.integers(0, 1 << 12, n).astype(np.uint32)
    ea.copy_(dev_words(O.fle_encode(a2, 12)))
    for t in (bm_pred, bm_chain, bm_tree, bm_pages, outs[0], pages[0][2][0]):
        t.zero_()
    g.replay()
    torch.cuda.synchronize()
    W = (n + 63) // 64
    assert np.array_equal(bits_of(words(bm_pred[:W]), n), a2 < 1000)
    assert np.array_equal(bits_of(words(outs[0][:W]), n), a2 < 1000)
    assert np.array_equal(bits_of(words(pages[0][2][0][:W]), n), a2 >= 2000)
    assert np.array_equal(bits_of(words(bm_chain[:W]), n), (a2 >= 100) & (a2 < 3000) & (b < 40))
    assert np.array_equal(bits_of(words(bm_pages[:W]), n), (a2 >= 100) & (a2 < 3000) & (b < 40))
    assert np.array_equal(bits_of(words(bm_tree[:W]), n),
                          ((a2 < 500) & (b >= 10)) | ((a2 >= 3500) & (b < 5)))


def test_nan_has_no_dictionary_order(capi, O):
    """FLOAT / DOUBLE dictionaries are ordered by operator< (dict-encoding.h:370-372, quirk Q16):
    a NaN gives std::sort / lower_bound no strict weak order.  The build refuses such dictionaries
    (the writer falls back to PLAIN, as for more than 40000 entries) and a NaN literal selects no row."""
    for t, npt, tt in ((O.T_FLOAT, np.float32, torch.float32), (O.T_DOUBLE, np.float64, torch.float64)):
        vals = np.array([1.5, np.nan, -2.0, 1.5, 7.25, np.nan, 0.0] * 50, dtype=npt)
        with pytest.raises(capi.IpsError) as ei:
            capi.dict_encode(torch.from_numpy(vals).cuda(), t)
        assert ei.value.status == 2          # IPS_ERR_UNSUPPORTED: use PLAIN
        page = np.sort(np.array([-2.0, 0.0, 1.5, 7.25], dtype=npt))
        with pytest.raises(capi.IpsError):   # a dictionary page holding a NaN is not ascending
            capi.Dict(np.append(page, npt(np.nan)).view(np.uint8), t)
        # the same column without the NaNs encodes; NaN literals translate to "no row"
        clean = vals[~np.isnan(vals)]
        pg, bw, enc = capi.dict_encode(torch.from_numpy(clean).cuda(), t)
        assert np.array_equal(pg.view(npt), page) and bw == 2
        dd = capi.Dict(pg, t)
        for op in (O.OP_EQ, O.OP_LT, O.OP_LE, O.OP_GT, O.OP_GE):
            assert dd.translate(op, npt(np.nan))[0] == capi.XL_ALL_FALSE
            assert not words(dd.pred(enc, len(clean), bw, op, npt(np.nan))).any()
        kind, fle_op, codes = dd.translate(O.OP_IN, np.array([np.nan, 1.5, np.nan], dtype=npt))
        assert kind == capi.XL_FLE and codes == [2]
        dd.close()
    # the plain-select streaming path keeps NaN payloads bit-exact (they are just slots)
    page64 = torch.from_numpy(np.tile(np.array([np.nan, 1.0, -0.0, np.inf]), 4096)).cuda()
    bm = torch.full((page64.numel() // 64,), -1, dtype=torch.int64, device="cuda")
    bv, cnt = capi.plain_select(page64.view(torch.int64), page64.numel(), O.T_DOUBLE, bm)
    assert torch.equal(capi.batches_compact(bv, cnt, page64.numel()), page64.view(torch.int64))


@pytest.mark.parametrize("type_name", TYPES)
def test_plain_select(capi, O, type_name):
    """Late materialisation on PLAIN pages: the slots of the rows a bitmap selects, per batch, in
    row order (== the reference's Decode(.., skip_rows) walk over the skip list)."""
    t = getattr(O, type_name)
    rng = np.random.default_rng(500 + t)
    for n in (1, 31, 2048, 2049, 70003):
        npt = O.NP_TYPES[t]
        vals = (rng.integers(-100, 100, n).astype(npt) if np.issubdtype(npt, np.integer)
                else rng.normal(0, 50, n).astype(npt))
        page = O.plain_encode(vals, t)
        d_page = torch.from_numpy(np.concatenate([page, np.zeros(16, np.uint8)])).cuda()
        for sel_p in (0.0, 0.03, 0.5, 1.0):
            sel = rng.random(n) < sel_p
            bm_words = np.zeros((n + 63) // 64 + 2, np.uint64)
            idx = np.nonzero(sel)[0]
            np.bitwise_or.at(bm_words, idx >> 6, np.uint64(1) << (idx & 63).astype(np.uint64))
            bm_words[-2:] = np.uint64(0xFFFFFFFFFFFFFFFF)            # dirt behind the last word
            if n % 64:
                bm_words[(n - 1) >> 6] |= ~np.uint64(0) << np.uint64(n % 64)   # dirt beyond n_rows
            bm = torch.from_numpy(bm_words.view(np.int64)).cuda()
            bvals, counts = capi.plain_select(d_page, n, t, bm)
            c = counts.cpu().numpy()
            slots = bvals.cpu().numpy()
            got = np.concatenate([slots[b * 2048: b * 2048 + c[b]] for b in range(len(c))])
            stride = len(page) // n
            exp = page.view({4: np.int32, 8: np.int64}[stride])[sel]
            assert np.array_equal(got, exp), (type_name, n, sel_p)
            assert c.sum() == sel.sum()


@pytest.mark.parametrize("type_name", TYPES)
def test_plain_scan(capi, O, type_name):
    """Fused PLAIN scan == ips_plain_pred (oracle-checked) + ips_plain_select, for single
    comparisons, a BETWEEN pair and an IN list, under both operand-order semantics."""
    t = getattr(O, type_name)
    npt = O.NP_TYPES[t]
    rng = np.random.default_rng(640 + t)
    for n in (1, 63, 2048, 2049, 70003):
        vals = (rng.integers(-100, 100, n).astype(npt) if np.issubdtype(npt, np.integer)
                else rng.normal(0, 50, n).astype(npt))
        page = O.plain_encode(vals, t)
        d_page = torch.from_numpy(np.concatenate([page, np.zeros(16, np.uint8)])).cuda()
        stride = len(page) // n
        slots = page.view({4: np.int32, 8: np.int64}[stride])

        def check(bitmap, bvals, counts, sel):
            assert np.array_equal(bits_of(words(bitmap), n), sel)
            c = counts.cpu().numpy()
            s_ = bvals.cpu().numpy()
            got = np.concatenate([s_[b * 2048: b * 2048 + c[b]] for b in range(len(c))])
            assert np.array_equal(got, slots[sel])

        lit = vals[n // 2]
        for sem in (O.SEM_SQL, O.SEM_REFERENCE):
            for op in (O.OP_EQ, O.OP_LT, O.OP_LE, O.OP_GT, O.OP_GE):
                ref = bits_of(O.plain_pred(page, n, t, op, lit, sem), n)
                check(*capi.plain_scan(d_page, n, t, op, lit, semantics=sem), ref)
        lo, hi = np.sort(rng.choice(vals, 2))
        sel = (vals >= lo) & (vals <= hi)
        check(*capi.plain_scan(d_page, n, t, O.OP_GE, lo, op2=O.OP_LE, literal2=hi), sel)
        # every row / no row selected: the index list is worked off in windows of 512 entries
        check(*capi.plain_scan(d_page, n, t, O.OP_GE, vals.min()), np.ones(n, bool))
        check(*capi.plain_scan(d_page, n, t, O.OP_LT, vals.min()), np.zeros(n, bool))
        lst = rng.choice(vals, min(5, n))
        check(*capi.plain_scan(d_page, n, t, O.OP_IN, lst), np.isin(vals, lst))
    with pytest.raises(capi.IpsError):
        capi.plain_scan(d_page, n, t, O.OP_IN, lst, semantics=O.SEM_REFERENCE)


def test_between_on_a_32_bit_column(capi, O):
    """Two comparisons on one w=32 column in one pass (the early-pruning kernel's pair form):
    constants that are present, rows that share the constants' high 16 bits (undecided after the
    high planes), and sub-tiles with and without such rows."""
    rng = np.random.default_rng(3232)
    n = 9 * 2048 + 77
    vals = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    lo, hi = np.uint32(0x40001234), np.uint32(0xC000BEEF)
    vals[5000:5100] = (lo & np.uint32(0xFFFF0000)) | rng.integers(0, 1 << 16, 100).astype(np.uint32)
    vals[12000:12003] = [lo, hi, hi]
    vals[-3:] = (hi & np.uint32(0xFFFF0000)) | np.array([0, 0xBEEF, 0xFFFF], np.uint32)
    enc = dev_words(O.fle_encode(vals, 32))
    cols = [capi.fle_column(enc, 32)]
    L, AND, OR = capi.leaf, capi.and_node, capi.or_node
    for ops, joiner, truth in (
            ((O.OP_GE, O.OP_LE), AND, (vals >= lo) & (vals <= hi)),
            ((O.OP_GT, O.OP_LT), AND, (vals > lo) & (vals < hi)),
            ((O.OP_LT, O.OP_GT), OR, (vals < lo) | (vals > hi)),
            ((O.OP_EQ, O.OP_EQ), OR, (vals == lo) | (vals == hi))):
        nodes = [L(0, ops[0], int(lo)), L(0, ops[1], int(hi)), joiner()]
        got = bits_of(words(capi.eval_program(nodes, cols, n)), n)
        assert np.array_equal(got, truth), ops


@pytest.mark.parametrize("type_name,D,bw", [("T_INT32", 12000, 14), ("T_INT32", 20000, 15), ("T_INT32", 30000, 15),
                                            ("T_INT32", 35000, 16), ("T_INT64", 6000, 13), ("T_INT64", 15000, 14), ("T_INT64", 17000, 15), ("T_INT32", 40000, 16), ("T_INT64", 40000, 16)])
def test_dictionary_decode_with_a_shared_lds_dictionary(capi, O, type_name, D, bw):
    """DictDecoder::GetValue x n for dictionaries larger than the 32 KiB a workgroup copies for itself:
    from 2^20 rows on, one workgroup of 16 / 8 / 4 waves per CU shares a copy in dynamic LDS when it fits
    (the largest, D = 40000, keep the entries that do not fit in L2); ragged size, every entry hit,
    bad_index stays 0."""
    t = getattr(O, type_name)
    npt = O.NP_TYPES[t]
    rng = np.random.default_rng(D)
    n = (1 << 20) + 12345
    entries = np.sort(rng.choice(np.arange(-10 ** 9, 10 ** 9, 3), D, replace=False)).astype(npt)
    codes = rng.integers(0, D, n).astype(np.uint32)
    codes[:D] = np.arange(D, dtype=np.uint32)
    dd = capi.Dict(entries.view(np.uint8), t)
    enc = dev_words(O.fle_encode(codes, bw))
    out, bad = dd.decode(enc, n, bw)
    assert int(bad.item()) == 0
    assert np.array_equal(out.cpu().numpy().astype(npt), entries[codes])
    dd.close()


@pytest.mark.parametrize("type_name", ["T_INT32", "T_INT64"])
@pytest.mark.parametrize("D,bw", [(200, 8), (1000, 10), (4096, 12), (9000, 14), (20000, 15)])
def test_dictionary_scan_dense_selections(capi, O, type_name, D, bw):
    """ips_dict_scan / ips_dict_select on sub-tiles where most or all rows are selected (the dense
    paths: four gathered entries per lane and 16-byte store while the dictionary sits in LDS or L1,
    one entry per lane beyond), against numpy; a code outside the dictionary raises bad_index."""
    t = getattr(O, type_name)
    npt = O.NP_TYPES[t]
    rng = np.random.default_rng(D + bw)
    n = 30011
    entries = np.sort(rng.choice(np.arange(-10 ** 8, 10 ** 8, 3), D, replace=False)).astype(npt)
    codes = rng.integers(0, D, n).astype(np.uint32)
    dd = capi.Dict(entries.view(np.uint8), t)
    enc = dev_words(O.fle_encode(codes, bw))
    vals = entries[codes]
    for frac in (0.5, 0.9, 1.0):
        lit = entries[min(int(frac * D), D - 1)]
        op, keep = (capi.OP_LE, vals <= entries[-1]) if frac >= 1.0 else (capi.OP_LT, vals < lit)
        bitmap, bvals, counts = dd.scan(enc, n, bw, op, np.array([entries[-1] if frac >= 1.0 else lit], dtype=npt))
        c = counts.cpu().numpy()
        bv = bvals.cpu().numpy()
        dense = np.concatenate([bv[b * 2048: b * 2048 + c[b]] for b in range(len(c))])
        assert np.array_equal(dense.astype(npt), vals[keep]), (D, bw, frac)
        sv, sc = dd.select(enc, n, bw, bitmap)
        scn, svn = sc.cpu().numpy(), sv.cpu().numpy()
        dense2 = np.concatenate([svn[b * 2048: b * 2048 + scn[b]] for b in range(len(scn))])
        assert np.array_equal(dense2.astype(npt), vals[keep]), (D, bw, frac)
    dd.close()


def test_in_sets_of_any_length(capi, O):
    """IN lists beyond the 256 constants a kernel argument holds (FleDecoder::In / DictDecoder::In take
    a vector of any length, fle-encoding.h:8236-8313, dict-encoding.h:523-541): K = 40000 on a w = 16
    dictionary column (the reference's dictionary cap) through the predicate, the fused scan + gather
    and a program leaf (REQUIRED and OPTIONAL), K = 3000 on raw FLE columns of 12 and 21 bits, an
    empty set, members beyond the column's width -- against the oracle's In on the same list."""
    rng = np.random.default_rng(77)
    n = 150001
    D = 40000                                             # the reference's dictionary cap (dict-encoding.h:157)
    dict_vals = np.sort(rng.choice(np.arange(-10 ** 6, 10 ** 6), D, replace=False)).astype(np.int32)
    codes = rng.integers(0, D, n).astype(np.uint32)
    enc_h = O.fle_encode(codes, 16)
    enc = dev_words(enc_h)
    dd = capi.Dict(dict_vals.view(np.uint8), capi.T_INT32)
    members = rng.choice(D, 39000, replace=False)
    lits = np.concatenate([dict_vals[members], dict_vals[members[:500]], np.arange(500, dtype=np.int32) + 10 ** 7])
    lits = lits.astype(np.int32)                          # 40000 literals: duplicates and ones that are no entries
    s = capi.InSet(lits, dict_=dd)
    assert len(lits) == 40000 and s.size == 39000
    truth = np.isin(codes, members)
    ref = O.fle_pred(enc_h, n, 16, O.OP_IN, np.sort(members).tolist())          # the oracle's In over 40000 codes
    assert np.array_equal(bits_of(ref, n), truth)
    assert np.array_equal(words(capi.fle_pred_inset(enc, n, 16, s)), ref)
    bitmap, bvals, counts = capi.dict_scan_inset(dd, enc, n, 16, s)
    assert np.array_equal(words(bitmap), ref)
    assert np.array_equal(capi.batches_compact(bvals, counts, n).cpu().numpy(), dict_vals[codes[truth]])
    # a program leaf: (codes IN set) AND (other < 9), then on an OPTIONAL column
    other = rng.integers(0, 32, n).astype(np.uint32)
    enc_o = dev_words(O.fle_encode(other, 5))
    nodes = [capi.inset_leaf(0, s), capi.leaf(1, capi.OP_LT, 9), capi.and_node()]
    got = capi.eval_program(nodes, [capi.fle_column(enc, 16), capi.fle_column(enc_o, 5)], n)
    assert np.array_equal(bits_of(words(got), n), truth & (other < 9))
    is_set = rng.random(n) >= 0.25
    k = int(is_set.sum())
    defs = dev_words(O.fle_encode(is_set.astype(np.uint32), 1))
    enc_nn = dev_words(O.fle_encode(codes[is_set], 16))
    ncol = capi.nullable_fle_column(defs, 1, 1, enc_nn, 16, ((k + 63) // 64) * 64)
    got = capi.eval_program([capi.inset_leaf(0, s)], [ncol], n)
    assert np.array_equal(bits_of(words(got), n), truth & is_set)
    # the same leaf over a page list
    cut = [50001, 33, 70000, n - 50001 - 33 - 70000]
    pages, pos = [], 0
    for m in cut:
        pages.append((dev_words(O.fle_encode(codes[pos:pos + m], 16)), m, 16))
        pos += m
    chunk = capi.Chunk(pages)
    got = capi.eval_program_chunks([capi.inset_leaf(0, s)], [chunk])
    assert np.array_equal(words(got), ref)
    chunk.close()
    s.close()
    dd.close()
    # raw FLE columns: 12 bits (table) and 21 bits (member by member), members beyond the width ignored
    for bw, K in ((12, 3000), (21, 600)):
        vals = rng.integers(0, 1 << bw, 40003, dtype=np.uint64).astype(np.uint32)
        e_h = O.fle_encode(vals, bw)
        e = dev_words(e_h)
        mem = rng.choice(1 << bw, K, replace=False).astype(np.uint64)
        s = capi.InSet(np.concatenate([mem, np.array([1 << bw, (1 << 31) + 5, 1 << 40], np.uint64)]))
        ref = O.fle_pred(e_h, len(vals), bw, O.OP_IN, np.sort(mem).tolist())
        assert np.array_equal(words(capi.fle_pred_inset(e, len(vals), bw, s)), ref), bw
        bitmap, bvals, counts = capi.fle_scan_inset(e, len(vals), bw, s)
        assert np.array_equal(words(bitmap), ref), bw
        assert np.array_equal(capi.batches_compact(bvals, counts, len(vals)).cpu().numpy().view(np.uint32),
                              vals[np.isin(vals, mem.astype(np.uint32))]), bw
        s.close()
    empty = capi.InSet(np.zeros(0, np.uint64))
    assert int(capi.fle_pred_inset(enc, n, 16, empty).abs().sum().item()) == 0
    empty.close()


@pytest.mark.gpu
def test_one_pass_conjunct_chain(capi, O, request):
    """ips_set_program_strategy(ONE_PASS): left-deep AND / OR chains of 2..6 operands over REQUIRED FLE
    columns of every width -- single comparisons, pairs on one column (BETWEEN and its OR twin), IN lists --
    run as ONE launch whose per-operand code is selected by a switch on the width (ips_chain.hip).  Checked
    against numpy on the raw codes and against the per-operand plan, row counts around the 2048-row sub-tile."""
    request.addfinalizer(lambda: capi.set_program_strategy(capi.PROGRAM_AUTO))
    rng = np.random.default_rng(77)
    L, AND, OR = capi.leaf, capi.and_node, capi.or_node
    cmp_np = {O.OP_EQ: np.equal, O.OP_LT: np.less, O.OP_LE: np.less_equal, O.OP_GT: np.greater, O.OP_GE: np.greater_equal}
    widths_seen = set()
    for trial in range(40):
        n = int(rng.choice([1, 63, 2047, 2048, 2049, 6145, 40961, 100003]))
        n_ops = int(rng.integers(2, 7))
        # the first trials walk through every width once
        widths = [int(rng.integers(1, 33)) for _ in range(n_ops)]
        if trial < 16:
            widths[0] = 2 * trial + 1
            widths[1] = 2 * trial + 2
        widths_seen.update(widths)
        cols, raw, nodes, keep = [], [], [], []
        exp = None
        for i, w in enumerate(widths):
            hi = (1 << w) - 1
            # a narrow value range now and then so that EQ / IN select something on wide columns
            span = hi if rng.random() < 0.6 else min(hi, 15)
            v = rng.integers(0, span + 1, n, dtype=np.uint64).astype(np.uint32)
            raw.append(v)
            keep.append(dev_words(O.fle_encode(v, w)))  # (the column descriptor only holds the address)
            cols.append(capi.fle_column(keep[-1], w))
            kind = rng.integers(0, 3)
            if kind == 0:
                op = int(rng.integers(0, 5))
                c = int(rng.integers(0, span + 1))
                nodes.append(L(i, op, c))
                sel = cmp_np[op](v, np.uint32(c))
            elif kind == 1:
                op1, op2 = int(rng.integers(0, 5)), int(rng.integers(0, 5))
                c1, c2 = int(rng.integers(0, span + 1)), int(rng.integers(0, span + 1))
                nodes += [L(i, op1, c1), L(i, op2, c2)]
                s1, s2 = cmp_np[op1](v, np.uint32(c1)), cmp_np[op2](v, np.uint32(c2))
                if rng.random() < 0.7:
                    nodes.append(AND())
                    sel = s1 & s2
                else:
                    nodes.append(OR())
                    sel = s1 | s2
            else:
                k = int(rng.integers(1, 17))
                members = [int(x) for x in rng.integers(0, span + 1, k)]
                nodes.append(L(i, O.OP_IN, members))
                sel = np.isin(v, np.array(members, dtype=np.uint32))
            if i == 0:
                exp = sel
            elif rng.random() < 0.75:
                nodes.append(AND())
                exp = exp & sel
            else:
                nodes.append(OR())
                exp = exp | sel
        capi.set_program_strategy(capi.PROGRAM_ONE_PASS)
        got = bits_of(words(capi.eval_program(nodes, cols, n)), n)
        assert np.array_equal(got, exp), (trial, n, widths)
        capi.set_program_strategy(capi.PROGRAM_PER_OPERAND)
        ref = bits_of(words(capi.eval_program(nodes, cols, n)), n)
        assert np.array_equal(ref, exp), (trial, n, widths)
    assert widths_seen == set(range(1, 33))

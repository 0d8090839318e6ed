"""Predicates on OPTIONAL (nullable) columns through the fused leaf (ips_fle_pred_nullable /
ips_dict_pred_nullable) against the reference's three steps restated by the oracle
(hdfs-parquet-scanner.cc:338-345: fle_def_levels_->Eq, data predicate over bits.count() values,
IntersectBitset :326-331)."""
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


def oracle_leaf(O, defs, def_bw, max_def, n, data_enc, k, bw, op, consts):
    nonnull = O.fle_pred(defs, n, def_bw, O.OP_EQ, max_def)
    assert int(bits_of(nonnull, n).sum()) == k
    sub = O.fle_pred(data_enc, k, bw, op, consts) if k else np.zeros(1, np.uint64)
    return O.bitmap_expand(nonnull, sub, n)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 4097, 70001, 262144, 262145, 3 * 262144 + 77, 1 << 21])
@pytest.mark.parametrize("null_frac", [0.0, 0.1, 0.5, 0.97, 1.0])
def test_nullable_leaf_vs_oracle(capi, O, n, null_frac):
    rng = np.random.default_rng(n * 7 + int(null_frac * 100))
    bw = int(rng.integers(1, 17))
    is_set = rng.random(n) >= null_frac
    if null_frac == 0.0:
        is_set[:] = True
    k = int(is_set.sum())
    vals = rng.integers(0, 1 << bw, max(k, 1)).astype(np.uint32)[:k]
    defs = O.fle_encode(is_set.astype(np.uint32), 1)
    enc = O.fle_encode(vals, bw) if k else np.zeros(2, np.uint64)
    d_defs, d_enc = dev_words(defs), dev_words(enc)
    n_data = ((k + 63) // 64) * 64          # what the host knows: blocks * 64
    c = int(rng.integers(0, 1 << bw))
    for op, consts in ((O.OP_LT, c), (O.OP_GE, c), (O.OP_EQ, int(vals[0]) if k else 0),
                       (O.OP_IN, [c, (c + 1) % (1 << bw), 0])):
        got = words(capi.fle_pred_nullable(d_defs, 1, 1, n, d_enc, n_data, bw, op, consts))
        exp = oracle_leaf(O, defs, 1, 1, n, enc, k, bw, op, consts)
        assert np.array_equal(got, exp), (n, null_frac, bw, op)
    # row-model check of one of them
    got = bits_of(words(capi.fle_pred_nullable(d_defs, 1, 1, n, d_enc, n_data, bw, O.OP_LT, c)), n)
    expect = np.zeros(n, bool)
    expect[np.flatnonzero(is_set)[vals < c]] = True
    assert np.array_equal(got, expect)


@pytest.mark.parametrize("bw", list(range(1, 33)))
def test_nullable_leaf_every_width(capi, O, bw):
    """Every code width through the one-pass leaf (fle_leaf_kernel: comparisons, short IN lists,
    pairs via the program) and through the routes it hands back (long IN lists -> membership table,
    w = 32 comparisons -> early-pruning predicate + expand); ragged row count, a data buffer that is
    not a whole number of blocks, output poisoned first."""
    rng = np.random.default_rng(1000 + bw)
    n = 150001 + 64 * bw
    is_set = rng.random(n) >= 0.3
    is_set[70000:90000] = False            # a stretch of NULLs wider than a wave's 16384 rows
    is_set[100000:120000] = True
    k = int(is_set.sum())
    hi = (1 << bw) - 1
    vals = rng.integers(0, hi + 1, k, dtype=np.uint64).astype(np.uint32)
    c = int(vals[k // 3])
    if bw == 32:   # stretches of rows that stay undecided after the high 16 planes (early pruning)
        vals[1000:6000] = (c & 0xFFFF0000) | (vals[1000:6000] & 0xFFFF)
        vals[50000:50003] = c
    defs, enc = O.fle_encode(is_set.astype(np.uint32), 1), O.fle_encode(vals, bw)
    d_defs, d_enc = dev_words(defs), dev_words(enc)
    long_list = [int(v) for v in vals[:40]]
    out = torch.empty((n + 63) // 64, dtype=torch.int64, device="cuda")
    for op, consts in ((O.OP_LT, c), (O.OP_GE, c), (O.OP_EQ, c), (O.OP_IN, [c, 0, hi]), (O.OP_IN, long_list)):
        out.fill_(-1)
        capi.fle_pred_nullable(d_defs, 1, 1, n, d_enc, k, bw, op, consts, bitmap=out)
        exp = oracle_leaf(O, defs, 1, 1, n, enc, k, bw, op, consts)
        assert np.array_equal(words(out), exp), (bw, op)
    # BETWEEN in one pass, AND-ed / OR-ed into a bitmap of another column (combine modes)
    other = rng.integers(0, 16, n).astype(np.uint32)
    d_other = dev_words(O.fle_encode(other, 4))   # (the column descriptor holds the pointer only)
    cols = [capi.fle_column(d_other, 4), capi.nullable_fle_column(d_defs, 1, 1, d_enc, bw, k)]
    full = np.full(n, -1, np.int64)
    full[is_set] = vals
    lo, up = sorted((c, int(vals[k // 2])))
    between = (full >= lo) & (full <= up)
    L, AND, OR = capi.leaf, capi.and_node, capi.or_node
    got = bits_of(words(capi.eval_program([L(0, O.OP_LT, 9), L(1, O.OP_GE, lo), L(1, O.OP_LE, up), AND(), AND()], cols, n)), n)
    assert np.array_equal(got, (other < 9) & between)
    got = bits_of(words(capi.eval_program([L(0, O.OP_LT, 3), L(1, O.OP_GE, lo), L(1, O.OP_LE, up), AND(), OR()], cols, n)), n)
    assert np.array_equal(got, (other < 3) | between)


def test_nullable_leaf_wider_definition_levels(capi, O):
    """max_def_level 2 (a nested OPTIONAL): levels are 2 bits wide, NOT NULL <=> level == 2."""
    rng = np.random.default_rng(5)
    n, bw = 100003, 9
    levels = rng.integers(0, 3, n).astype(np.uint32)
    is_set = levels == 2
    k = int(is_set.sum())
    vals = rng.integers(0, 1 << bw, k).astype(np.uint32)
    defs, enc = O.fle_encode(levels, 2), O.fle_encode(vals, bw)
    got = words(capi.fle_pred_nullable(dev_words(defs), 2, 2, n, dev_words(enc), k, bw, O.OP_GT, 300))
    assert np.array_equal(got, oracle_leaf(O, defs, 2, 2, n, enc, k, bw, O.OP_GT, 300))


def test_nullable_leaf_constants_outside_the_domain_and_short_data(capi, O):
    rng = np.random.default_rng(6)
    n, bw = 50000, 6
    is_set = rng.random(n) < 0.7
    k = int(is_set.sum())
    vals = rng.integers(0, 1 << bw, k).astype(np.uint32)
    defs, enc = O.fle_encode(is_set.astype(np.uint32), 1), O.fle_encode(vals, bw)
    d_defs, d_enc = dev_words(defs), dev_words(enc)
    nonnull = O.fle_pred(defs, n, 1, O.OP_EQ, 1)
    # LT 2^w: every NON-NULL row; GT 2^w: none (quirk Q6 stance: unsigned SQL meaning)
    assert np.array_equal(words(capi.fle_pred_nullable(d_defs, 1, 1, n, d_enc, k, bw, O.OP_LT, 1 << bw)), nonnull)
    assert not words(capi.fle_pred_nullable(d_defs, 1, 1, n, d_enc, k, bw, O.OP_GT, 1 << bw)).any()
    # a data buffer shorter than the NOT-NULL count: the rows beyond it are not selected
    short = (k // 2 // 64) * 64
    got = bits_of(words(capi.fle_pred_nullable(d_defs, 1, 1, n, d_enc, short, bw, O.OP_GE, 0)), n)
    expect = np.zeros(n, bool)
    expect[np.flatnonzero(is_set)[:short]] = True
    assert np.array_equal(got, expect)
    with pytest.raises(capi.IpsError):
        capi.fle_pred_nullable(d_defs, 1, 3, n, d_enc, k, bw, O.OP_LT, 3)   # max_def 3 in 1 bit


@pytest.mark.parametrize("type_name", ["T_INT32", "T_INT64", "T_DOUBLE"])
def test_nullable_dictionary_leaf(capi, O, type_name):
    """ColumnReader<T>::Lt/Ge/In on an OPTIONAL dictionary page, literals translated to codes."""
    t = getattr(O, type_name)
    npt = O.NP_TYPES[t]
    rng = np.random.default_rng(40 + t)
    n = 300007
    is_set = rng.random(n) < 0.85
    k = int(is_set.sum())
    pool = np.unique(rng.integers(-5000, 5000, 700)).astype(npt)
    data_vals = pool[rng.integers(0, len(pool), k)]
    defs = O.fle_encode(is_set.astype(np.uint32), 1)
    d, dict_page, codes = O.dict_build(data_vals, t)
    page = O.dict_write_data(codes, len(d))
    bw, blocks = int(page[0]), np.frombuffer(page[1:].tobytes(), dtype=np.uint64)
    dd = capi.Dict(dict_page, t)
    d_defs, d_blocks = dev_words(defs), dev_words(blocks)
    n_data = (len(blocks) // bw) * 64
    lit = npt(pool[len(pool) // 3])
    cases = [(O.OP_LT, lit, data_vals < lit), (O.OP_GE, lit, data_vals >= lit),
             (O.OP_EQ, lit, data_vals == lit), (O.OP_LT, npt(10 ** 6), np.ones(k, bool)),
             (O.OP_GT, npt(10 ** 6), np.zeros(k, bool)),
             (O.OP_IN, np.array([pool[1], pool[5], 777777], npt), np.isin(data_vals, [pool[1], pool[5]]))]
    for op, lits, data_truth in cases:
        got = bits_of(words(dd.pred_nullable(d_defs, 1, 1, n, d_blocks, n_data, bw, op, lits)), n)
        expect = np.zeros(n, bool)
        expect[np.flatnonzero(is_set)[data_truth]] = True
        assert np.array_equal(got, expect), (type_name, op)
    dd.close()


def test_nullable_leaf_2p28_rows_properties(capi, ips):
    """Full size (2^28 rows, w = 12, 10 % NULLs): against torch on the raw columns."""
    n, bw = 1 << 28, 12
    dev = torch.device("cuda")
    nn = capi.synth_u32(0x5EED0D1, n, 32)
    is_set = (nn.to(torch.int64) & 0xFFFFFFFF) >= int(0.1 * (1 << 32))
    del nn
    defs = capi.fle_encode(is_set.to(torch.int32), 1)
    k = int(is_set.sum().item())
    vals = capi.synth_u32(0x5EED0D2, k, bw)
    enc = capi.fle_encode(vals, bw)
    ws = capi.nullable_workspace(n, dev)
    bm = capi.fle_pred_nullable(defs, 1, 1, n, enc, ((k + 63) // 64) * 64, bw, capi.OP_LT, 409, workspace=ws)
    sel_rows = torch.zeros(n, dtype=torch.bool, device=dev)
    sel_rows[is_set] = vals < 409
    assert capi.bitmap_count(bm, n) == int(sel_rows.sum().item())
    w = sel_rows.view(-1, 64).to(torch.int64)
    packed = (w << torch.arange(64, device=dev, dtype=torch.int64)).sum(dim=1)
    assert torch.equal(packed, bm)


def test_program_with_optional_columns(capi, O):
    """ips_eval_program over REQUIRED and OPTIONAL columns: leaves on an OPTIONAL column go through
    the nullable leaf (data predicate + expand, combined into the plan's bitmaps), a BETWEEN on it
    in one pass; NULL rows never pass a leaf (SQL: NULL compared with anything is not true)."""
    rng = np.random.default_rng(99)
    n = 300001
    a = rng.integers(0, 1 << 12, n).astype(np.uint32)                 # REQUIRED w=12
    b_set = rng.random(n) < 0.8                                        # OPTIONAL w=7
    b_vals = rng.integers(0, 1 << 7, int(b_set.sum())).astype(np.uint32)
    c_set = rng.random(n) < 0.5                                        # OPTIONAL w=3, levels 2 bits wide
    c_levels = np.where(c_set, 2, rng.integers(0, 2, n)).astype(np.uint32)
    c_vals = rng.integers(0, 1 << 3, int(c_set.sum())).astype(np.uint32)
    b_full = np.zeros(n, np.int64) - 1
    b_full[b_set] = b_vals
    c_full = np.zeros(n, np.int64) - 1
    c_full[c_set] = c_vals
    ea = dev_words(O.fle_encode(a, 12))
    eb, db = dev_words(O.fle_encode(b_vals, 7)), dev_words(O.fle_encode(b_set.astype(np.uint32), 1))
    ec, dc = dev_words(O.fle_encode(c_vals, 3)), dev_words(O.fle_encode(c_levels, 2))
    cols = [capi.fle_column(ea, 12),
            capi.nullable_fle_column(db, 1, 1, eb, 7, ((len(b_vals) + 63) // 64) * 64),
            capi.nullable_fle_column(dc, 2, 2, ec, 3, len(c_vals))]
    L, AND, OR = capi.leaf, capi.and_node, capi.or_node
    nn_b, nn_c = b_full >= 0, c_full >= 0
    cases = [
        ([L(1, O.OP_LT, 40)], nn_b & (b_full < 40)),
        ([L(1, O.OP_GE, 20), L(1, O.OP_LE, 90), AND()], nn_b & (b_full >= 20) & (b_full <= 90)),
        ([L(0, O.OP_LT, 2000), L(1, O.OP_GE, 64), AND(), L(2, O.OP_EQ, 5), AND()],
         (a < 2000) & nn_b & (b_full >= 64) & nn_c & (c_full == 5)),
        ([L(0, O.OP_LT, 300), L(1, O.OP_IN, [1, 2, 3, 100]), OR(), L(2, O.OP_GT, 3), L(0, O.OP_GE, 4000), AND(), OR()],
         (a < 300) | (nn_b & np.isin(b_full, [1, 2, 3, 100])) | (nn_c & (c_full > 3) & (a >= 4000))),
    ]
    for nodes, truth in cases:
        assert capi.program_workspace_bytes(nodes, cols, n) > 0
        got = bits_of(words(capi.eval_program(nodes, cols, n)), n)
        assert np.array_equal(got, truth)
    # the leaf alone agrees with the stand-alone entry point
    solo = capi.eval_program([L(2, O.OP_LE, 4)], cols, n)
    assert torch.equal(solo, capi.fle_pred_nullable(dc, 2, 2, n, ec, len(c_vals), 3, O.OP_LE, 4))


@pytest.mark.parametrize("bw", [1, 3, 7, 8, 12, 16, 21, 32])
def test_select_nullable_against_row_model(capi, O, bw):
    """ips_dict_select_nullable (ReadValue(skip) over a whole selection on an OPTIONAL column,
    hdfs-parquet-scanner.cc:1006-1038): dense values of the selected NOT-NULL rows in row order,
    one NOT-NULL flag per selected row, both counts -- against a numpy row model; raw FLE values and
    dictionaries of 4- and 8-byte entries, ragged sizes, NULL stretches longer than a wave's 16 384
    rows, selectivities from a handful of rows to all, data buffers shorter than the NOT-NULL count,
    level widths 1 and 2."""
    rng = np.random.default_rng(500 + bw)
    for n, null_frac, sel_frac, max_def in ((1, 0.0, 1.0, 1), (65, 0.5, 0.5, 1), (4097, 0.1, 0.01, 1),
                                            (200003, 0.3, 0.1, 1), (200003 + 64 * bw, 0.1, 1.0, 1),
                                            (150001, 0.6, 0.3, 2), (70001, 1.0, 0.5, 1), (70001, 0.2, 0.0, 1)):
        is_set = rng.random(n) >= null_frac
        if n > 100000:
            is_set[30000:50000] = False
            is_set[60000:80000] = True
        sel = rng.random(n) < sel_frac
        if sel_frac >= 1.0:
            sel[:] = True
        k = int(is_set.sum())
        def_bw = 1 if max_def == 1 else 2
        levels = np.where(is_set, max_def, rng.integers(0, max_def, n)).astype(np.uint32)
        d_defs = dev_words(O.fle_encode(levels, def_bw))
        d_sel = dev_words(np.packbits(np.concatenate([sel, np.zeros((-n) % 64, bool)]), bitorder="little").view(np.uint64))
        for kind in ("raw", "i32", "i64"):
            if kind != "raw" and bw > 16:
                continue
            if kind == "raw":
                dd, codes = None, rng.integers(0, 1 << bw, max(k, 1), dtype=np.uint64).astype(np.uint32)[:k]
                values = codes.astype(np.int64)
            else:
                D = min(1 << bw, 3000)
                npt = np.int32 if kind == "i32" else np.int64
                entries = np.sort(rng.choice(np.arange(-10 ** 6, 10 ** 6), D, replace=False)).astype(npt)
                dd = capi.Dict(entries.view(np.uint8), capi.T_INT32 if kind == "i32" else capi.T_INT64)
                codes = rng.integers(0, D, max(k, 1)).astype(np.uint32)[:k]
                values = entries[codes].astype(np.int64)
            d_enc = dev_words(O.fle_encode(codes, bw)) if k else dev_words(np.zeros(2, np.uint64))
            for n_data in sorted({k, ((k + 63) // 64) * 64, k // 2}):
                usable = min(n_data, k)
                rank = np.cumsum(is_set) - 1                       # data row of a NOT-NULL row
                take = sel & is_set & (rank < usable)
                exp_dense = values[rank[take]]
                exp_flags = is_set[sel]
                dense, flags, n_sel, n_val = capi.select_nullable(dd, d_defs, def_bw, max_def, n, d_enc, n_data, bw, d_sel)
                ctx = (bw, n, null_frac, sel_frac, kind, n_data)
                assert n_sel == int(sel.sum()), ctx
                assert n_val == len(exp_dense), ctx
                got = dense.cpu().numpy()
                got = got.view(np.uint32).astype(np.int64) if kind == "raw" else got.astype(np.int64)
                assert np.array_equal(got, exp_dense), ctx
                got_flags = bits_of(words(flags), n_sel)
                assert np.array_equal(got_flags, exp_flags), ctx
            if dd is not None:
                dd.close()


@pytest.mark.parametrize("type_name", ["T_INT32", "T_INT64", "T_DOUBLE", "T_INT16"])
def test_optional_plain_page_sql_semantics(capi, O, type_name):
    """NULL-aware PLAIN pages (an OPTIONAL column that fell back from dictionary to PLAIN pages past
    40000 entries, dict-encoding.h:157).  The reference's PLAIN branch ignores the levels
    (hdfs-parquet-scanner.cc:346-348, quirk Q3); under SQL semantics a NULL row fails and the stored
    values are compared: expected = the oracle's PLAIN predicate over the stored values, expanded into
    the NOT-NULL positions (IntersectBitset, :326-331) -- alone and as a program leaf next to a
    REQUIRED FLE column."""
    t = getattr(O, type_name)
    npt = O.NP_TYPES[t]
    rng = np.random.default_rng(400 + t)
    for n in (1, 65, 4099, 300001):
        for null_frac in (0.0, 0.2, 0.9):
            is_set = rng.random(n) >= null_frac
            k = int(is_set.sum())
            vals = (rng.normal(0, 50, k) if t == O.T_DOUBLE else rng.integers(-300, 300, k)).astype(npt)
            defs_h = O.fle_encode(is_set.astype(np.uint32), 1)
            page_h = O.plain_encode(vals, t) if k else np.zeros(16, np.uint8)
            defs = dev_words(defs_h)
            page = torch.from_numpy(np.concatenate([page_h, np.zeros((-len(page_h)) % 16 + 16, np.uint8)])).cuda()
            nonnull = O.fle_pred(defs_h, n, 1, O.OP_EQ, 1)
            for op, lit in ((O.OP_LT, npt(17)), (O.OP_GE, npt(-40)), (O.OP_EQ, vals[0] if k else npt(0)),
                            (O.OP_IN, np.array([1, 2, 250], npt))):
                sub = O.plain_pred(page_h, k, t, op, lit, O.SEM_SQL) if k else np.zeros(1, np.uint64)
                exp = O.bitmap_expand(nonnull, sub, n)
                got = capi.plain_pred_nullable(defs, 1, 1, n, page, k, t, op, lit)
                assert np.array_equal(words(got), exp), (type_name, n, null_frac, op)
            # program: (plain_col < 17) AND (fle_col >= 3), then OR-ed the other way round
            other = rng.integers(0, 16, n).astype(np.uint32)
            enc_o = dev_words(O.fle_encode(other, 4))
            cols = [capi.nullable_plain_column(defs, 1, 1, page, t, k), capi.fle_column(enc_o, 4)]
            row_vals = np.zeros(n, npt)
            row_vals[is_set] = vals
            lt = is_set & (row_vals < npt(17))
            nodes = [capi.plain_leaf(0, O.OP_LT, npt(17), t), capi.leaf(1, O.OP_GE, 3), capi.and_node()]
            got = capi.eval_program(nodes, cols, n)
            assert np.array_equal(bits_of(words(got), n), lt & (other >= 3)), (type_name, n, null_frac)
            nodes = [capi.leaf(1, O.OP_LT, 2), capi.plain_leaf(0, O.OP_LT, npt(17), t), capi.or_node()]
            got = capi.eval_program(nodes, cols, n)
            assert np.array_equal(bits_of(words(got), n), lt | (other < 2)), (type_name, n, null_frac)

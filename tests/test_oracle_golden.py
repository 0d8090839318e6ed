"""CPU tests: pin the oracle (oracle/fle_oracle.c) against the reference's own test data
(tests/golden/reference_vectors.json: fle-test.cc / dict-test.cc / SURVEY known answers) and
against an independent row-at-a-time numpy model for the predicates, which the reference does not
test at all ("parity unpinned" by reference fixtures)."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "reference_vectors.json")))


def make_values(spec):
    if spec["pattern"] == "concat":
        return np.concatenate([np.full(cnt, v) for v, cnt in spec["parts"]]).astype(np.uint32)
    return (np.arange(spec["count"]) % spec["modulus"]).astype(np.uint32)


def bits_of(words, n):
    return np.unpackbits(np.ascontiguousarray(words).view(np.uint8), bitorder="little")[:n].astype(bool)


def model_pred(vals, op, consts, O):
    v = vals.astype(np.uint64)
    c = np.atleast_1d(consts).astype(np.uint64)
    return {O.OP_EQ: lambda: v == c[0], O.OP_LT: lambda: v < c[0], O.OP_LE: lambda: v <= c[0],
            O.OP_GT: lambda: v > c[0], O.OP_GE: lambda: v >= c[0],
            O.OP_IN: lambda: np.isin(v, c)}[op]()


# ---- fle-test.cc:202-242 ---------------------------------------------------------------------
@pytest.mark.parametrize("case", GOLD["specific_sequences"], ids=lambda c: c["name"])
def test_specific_sequences(O, case):
    vals = make_values(case["values"])
    bw = case["bit_width"]
    enc = O.fle_encode(vals, bw)
    assert enc.nbytes == case["encoded_len"] == O.fle_encoded_bytes(len(vals), bw)
    expected = np.array([int(h, 16) for h in case["expected_words_hex"]], dtype=np.uint64)
    # the reference pins only the defined (MSB-side) bits of the last block's words (quirk Q4)
    nblocks = (len(vals) + 63) // 64
    mask = np.full(len(enc), np.uint64(0xFFFFFFFFFFFFFFFF))
    rows_last = case["rows_in_last_block"]
    last_mask = np.uint64(((1 << rows_last) - 1) << (64 - rows_last))
    mask[(nblocks - 1) * bw:] = last_mask
    assert np.array_equal(enc & mask, expected & mask)
    # our encoder zero-pads, so here the match is exact
    assert np.array_equal(enc, expected)
    assert np.array_equal(O.fle_decode(enc, len(vals), bw), vals)


def test_specific_sequences_all_widths(O):
    for spec in (GOLD["specific_sequences"][0]["values"], GOLD["specific_sequences"][2]["values"]):
        vals = make_values(spec)
        for bw in range(1, 33):
            enc = O.fle_encode(vals, bw)
            assert enc.nbytes == 2 * bw * 8
            assert np.array_equal(O.fle_decode(enc, len(vals), bw), vals)


def test_survey_known_answer(O):
    ka = GOLD["survey_known_answer"]
    vals = make_values(ka["values"])
    enc = O.fle_encode(vals, ka["bit_width"])
    assert [int(h, 16) for h in ka["block0_words_hex"]] == [int(x) for x in enc]
    bm = O.fle_pred(enc, 64, ka["bit_width"], O.OP_LT, ka["lt_constant"])
    assert int(bm[0]) == int(ka["lt_bitmap_word_hex"], 16)


# ---- fle-test.cc:255-275 ---------------------------------------------------------------------
@pytest.mark.parametrize("bw", range(1, 33))
def test_round_trips(O, bw):
    rng = np.random.default_rng(1000 + bw)
    mod = 1 << bw
    cases = [np.arange(1) % mod, np.arange(1024) % mod, np.zeros(1024), np.ones(1024),
             rng.integers(0, mod, 1024)]
    for vals in cases:
        vals = vals.astype(np.uint32)
        enc = O.fle_encode(vals, bw)
        assert enc.nbytes == ((len(vals) + 63) // 64) * bw * 8
        assert np.array_equal(O.fle_decode(enc, len(vals), bw), vals)
        # stateful Get walks the same values and then reports end of data
        dec = O.FleDecoder(enc, enc.nbytes, bw)
        got = [dec.get() for _ in range(len(vals))]
        assert all(ok for ok, _ in got) and [v for _, v in got] == [int(x) for x in vals]


@pytest.mark.parametrize("bw", [1, 5, 8, 9, 16, 17, 31, 32])
def test_fast_unpack_equals_scalar(O, bw):
    rng = np.random.default_rng(7 + bw)
    vals = rng.integers(0, 1 << bw, 64 * 9, dtype=np.uint64).astype(np.uint32)
    enc = O.fle_encode(vals, bw)
    for b in range(9):
        assert np.array_equal(O.fast_unpack_block(enc[b * bw:(b + 1) * bw], bw),
                              vals[b * 64:(b + 1) * 64])


# ---- cursor semantics, fle-encoding.h:344-402 -----------------------------------
def test_cursor_semantics(O):
    bw, n = 11, 1000
    rng = np.random.default_rng(3)
    vals = rng.integers(0, 1 << bw, n).astype(np.uint32)
    enc = O.fle_encode(vals, bw)
    dec = O.FleDecoder(enc, enc.nbytes, bw)
    assert dec.get_skip(70) == (True, int(vals[70]))
    assert dec.get_skip(0) == (True, int(vals[71]))
    assert dec.skip(100)                       # cursor at 172
    assert dec.get() == (True, int(vals[172]))
    # predicates do not advance the cursor and start at it (mid-block start)
    bs = dec.pred(O.OP_LT, 300, 1000)
    assert np.array_equal(bs.bits(), vals[173:473] < 1000)
    assert dec.get() == (True, int(vals[173]))
    # end of data: Get returns false once the needed block starts at/after buffer+len
    dec2 = O.FleDecoder(enc, enc.nbytes, bw)
    assert dec2.get_skip(959)[0]               # last block (rows 960..1023 exist as a block)
    assert not dec2.get_skip(64)[0]


# ---- predicates vs the independent model ------------------------------------------------------
@pytest.mark.parametrize("bw", range(1, 33))
def test_predicates_vs_model(O, bw):
    rng = np.random.default_rng(bw)
    for n in (1, 63, 64, 65, 200, 3237):
        vals = rng.integers(0, 1 << bw, n, dtype=np.uint64).astype(np.uint32)
        enc = O.fle_encode(vals, bw)
        consts = {0, (1 << bw) - 1, int(vals[0]), int(vals[n // 2]), (1 << bw) // 10}
        for c in consts:
            for op in (O.OP_EQ, O.OP_LT, O.OP_LE, O.OP_GT, O.OP_GE):
                got = bits_of(O.fle_pred(enc, n, bw, op, c), n)
                assert np.array_equal(got, model_pred(vals, op, c, O)), (bw, n, op, c)
        lst = [int(vals[0]), int(vals[-1]), 0, (1 << bw) - 1]
        got = bits_of(O.fle_pred(enc, n, bw, O.OP_IN, lst), n)
        assert np.array_equal(got, model_pred(vals, O.OP_IN, lst, O))


# ---- dict-test.cc:118-147 + dict-encoding.h:461-541 ------------------------------------------
@pytest.mark.parametrize("type_name", ["T_INT8", "T_INT16", "T_INT32", "T_INT64", "T_FLOAT", "T_DOUBLE"])
def test_dict_numbers(O, type_name):
    t = getattr(O, type_name)
    for max_value, repeat in GOLD["dict_cases"]["cases"]:
        vals = np.repeat(np.arange(max_value), repeat).astype(O.NP_TYPES[t])
        d, page, codes = O.dict_build(vals, t)
        assert len(d) == len(set(vals.tolist()))            # num_entries == |set|
        assert np.all(d[:-1] < d[1:])                        # sorted ascending
        data = O.dict_write_data(codes, len(d))
        assert data[0] == O.bit_width_for_entries(len(d))
        ok, out = O.dict_decode(d, t, data, len(vals))
        assert ok and np.array_equal(out, vals)


def test_dict_survey_case(O):
    """5000 signed int32 / 300 distinct -> header byte 9, 5689 bytes."""
    rng = np.random.default_rng(9)
    pool = rng.choice(np.arange(-100000, 100000), 300, replace=False).astype(np.int32)
    vals = pool[rng.integers(0, 300, 5000)]
    vals[:300] = pool
    d, page, codes = O.dict_build(vals, O.T_INT32)
    data = O.dict_write_data(codes, len(d))
    assert len(d) == 300 and data[0] == 9 and len(data) == 1 + 79 * 9 * 8
    lits = [int(d[0]) - 5, int(d[0]), int(d[10]) + 1 if d[10] + 1 != d[11] else int(d[10]) - 1,
            int(d[150]), int(d[-1]), int(d[-1]) + 7]
    for lit in lits:
        for op, f in ((O.OP_EQ, np.equal), (O.OP_LT, np.less), (O.OP_LE, np.less_equal),
                      (O.OP_GT, np.greater), (O.OP_GE, np.greater_equal)):
            got = bits_of(O.dict_pred(d, O.T_INT32, data, 5000, op, lit), 5000)
            assert np.array_equal(got, f(vals, lit)), (op, lit)
    in_list = [int(d[3]), int(d[3]) + 1 if d[3] + 1 != d[4] else int(d[3]) - 1, int(d[200]), 10 ** 9]
    got = bits_of(O.dict_pred(d, O.T_INT32, data, 5000, O.OP_IN, in_list), 5000)
    assert np.array_equal(got, np.isin(vals, in_list))
    none = bits_of(O.dict_pred(d, O.T_INT32, data, 5000, O.OP_IN, [10 ** 9, -10 ** 9]), 5000)
    assert not none.any()


def test_bit_width_rule(O):
    # dict-encoding.h:76-80 with BitUtil::Log2 (ceil log2), bit-util.h:128-140
    assert [O.bit_width_for_entries(d) for d in (0, 1, 2, 3, 4, 5, 256, 257, 4096, 40000)] == \
        [0, 1, 1, 2, 2, 3, 8, 9, 12, 16]


# ---- PLAIN pages, parquet-common.h:197-250 ----------------------------------------------------
@pytest.mark.parametrize("type_name", ["T_INT8", "T_INT16", "T_INT32", "T_INT64", "T_FLOAT", "T_DOUBLE"])
def test_plain_pred(O, type_name):
    t = getattr(O, type_name)
    rng = np.random.default_rng(11)
    npt = O.NP_TYPES[t]
    if np.issubdtype(npt, np.integer):
        info = np.iinfo(npt)
        vals = rng.integers(max(info.min, -1000), min(info.max, 1000), 777).astype(npt)
    else:
        vals = rng.normal(0, 100, 777).astype(npt)
    page = O.plain_encode(vals, t)
    assert len(page) == 777 * O.lib().orc_plain_stride(t)
    lit = vals[5]
    fs = {O.OP_EQ: np.equal, O.OP_LT: np.less, O.OP_LE: np.less_equal, O.OP_GT: np.greater,
          O.OP_GE: np.greater_equal}
    for op, f in fs.items():
        sql = bits_of(O.plain_pred(page, 777, t, op, lit, O.SEM_SQL), 777)
        ref = bits_of(O.plain_pred(page, 777, t, op, lit, O.SEM_REFERENCE), 777)
        assert np.array_equal(sql, f(vals, lit))
        assert np.array_equal(ref, f(lit, vals))      # quirk Q1: literal OP x


# ---- scanner bitmap logic ---------------------------------------------------------------------
def test_expand_skiplist_select(O):
    rng = np.random.default_rng(5)
    n = 1500
    root_bits = rng.random(n) < 0.7
    root = np.packbits(root_bits, bitorder="little")
    root = np.concatenate([root, np.zeros(-len(root) % 8, np.uint8)]).view(np.uint64)
    k = int(root_bits.sum())
    sub_bits = rng.random(k) < 0.3
    sub = np.packbits(sub_bits, bitorder="little")
    sub = np.concatenate([sub, np.zeros(-len(sub) % 8 + 8, np.uint8)]).view(np.uint64)
    out = bits_of(O.bitmap_expand(root, sub, n), n)
    exp = np.zeros(n, bool)
    exp[np.flatnonzero(root_bits)[sub_bits]] = True       # IntersectBitset, scanner.cc:326-331
    assert np.array_equal(out, exp)

    bm = O.bitmap_expand(root, sub, n)
    skips, last = O.skip_list(bm, n)
    rows = np.flatnonzero(exp)
    assert np.array_equal(np.cumsum(skips + 1) - 1, rows)  # scanner.cc:1134-1148
    assert last == n - 1 - rows[-1]

    bw = 13
    vals = rng.integers(0, 1 << bw, n).astype(np.uint32)
    enc = O.fle_encode(vals, bw)
    assert np.array_equal(O.fle_select(enc, n, bw, bm), vals[exp])


def test_cpu_baseline_body(O):
    """The timed baseline produces the same bitmap / selected rows as the restatement."""
    bw, n = 32, 64 * 1024 + 37
    rng = np.random.default_rng(2)
    vals = rng.integers(0, 1 << bw, n, dtype=np.uint64).astype(np.uint32)
    enc = O.fle_encode(vals, bw)
    c = (1 << bw) // 10
    for mode in (0, 1):
        cnt, bm, sel = O.bench_fused(enc, n, bw, O.OP_LT, c, threads=3, mode=mode)
        assert np.array_equal(bm, O.fle_pred(enc, n, bw, O.OP_LT, c))
        assert cnt == int((vals < c).sum())

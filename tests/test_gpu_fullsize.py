"""GPU tests at BASELINE.json's full sizes.  2^20 rows (configs[0]/[1]) are compared bit-exactly
with the oracle; 2^28 rows use size-independent properties: encode->decode round trip,
popcount(bitmap) == sum(batch counts) == an independent torch count, compacted values == torch's
masked_select of the decoded column, idempotence, and select(scan's bitmap) == scan's batches."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def words(t):
    return t.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("bw", [32, 16, 8])
def test_config1_1m_rows_bit_exact(capi, ips, O, bw):
    n = 1 << 20
    vals = ips.synth.column_u32(ips.synth.SEED_HEADLINE, n, bw)
    c = ips.synth.lt_constant(bw)
    d_vals = capi.synth_u32(ips.synth.SEED_HEADLINE, n, bw)
    assert np.array_equal(d_vals.cpu().numpy().view(np.uint32), vals)   # generator twin
    enc = capi.fle_encode(d_vals, bw)
    enc_ref = O.fle_encode(vals, bw)
    assert np.array_equal(words(enc), enc_ref)
    bitmap, bvals, counts = capi.fle_scan(enc, n, bw, capi.OP_LT, c)
    bm_ref = O.fle_pred(enc_ref, n, bw, O.OP_LT, c)
    assert np.array_equal(words(bitmap), bm_ref)
    assert np.array_equal(words(capi.fle_pred(enc, n, bw, capi.OP_LT, c)), bm_ref)
    dense = capi.batches_compact(bvals, counts, n).cpu().numpy().view(np.uint32)
    assert np.array_equal(dense, O.fle_select(enc_ref, n, bw, bm_ref))
    assert abs(len(dense) / n - 0.10) < 0.005


def test_config2_2p28_rows_properties(capi, ips):
    n, bw = 1 << 28, 32
    c = ips.synth.lt_constant(bw)
    vals = capi.synth_u32(ips.synth.SEED_HEADLINE, n, bw)
    enc = capi.fle_encode(vals, bw)
    # encode -> decode round trip at full size
    dec = capi.fle_decode(enc, n, bw, 4)
    assert torch.equal(dec, vals)
    del dec
    outs = capi.alloc_scan_outputs(n, vals.device)
    bitmap, bvals, counts = capi.fle_scan(enc, n, bw, capi.OP_LT, c, outputs=outs)
    # independent answer from torch on the raw (unsigned) values
    mask = (vals.to(torch.int64) & 0xFFFFFFFF) < c
    n_sel = int(mask.sum().item())
    assert capi.bitmap_count(bitmap, n) == n_sel == int(counts.to(torch.int64).sum().item())
    assert abs(n_sel / n - 0.10) < 0.001
    dense = capi.batches_compact(bvals, counts, n)
    assert torch.equal(dense, torch.masked_select(vals, mask))
    # bitmap bits == mask (pack the mask the LSB-first way on the GPU)
    w = mask.view(-1, 64).to(torch.int64)
    packed = (w << torch.arange(64, device=w.device, dtype=torch.int64)).sum(dim=1)
    assert torch.equal(packed, bitmap)
    del mask, w, packed, dense
    # idempotence + predicate-only kernel agrees with the fused one
    bm2 = capi.fle_pred(enc, n, bw, capi.OP_LT, c)
    assert torch.equal(bm2, bitmap)
    # late materialisation against the given bitmap reproduces the fused batches
    b2, c2 = capi.fle_select(enc, n, bw, bm2)
    assert torch.equal(c2, counts)
    assert torch.equal(capi.batches_compact(b2, c2, n), capi.batches_compact(bvals, counts, n))


def test_int64_dictionary_between_selectivity_sweep(capi, ips, O):
    """configs[2]: int64 dictionary column (D = 4096, w = 12), BETWEEN = And(Ge lo, Le hi) at
    1 / 10 / 50 / 100 % selectivity, fused program vs numpy (And(Gt a, Lt b): the tests below)."""
    n, D = 1 << 22, 4096
    rng = np.random.default_rng(3)
    dict_vals = np.sort(rng.choice(np.arange(-2 ** 40, 2 ** 40, 2 ** 20), D, replace=False)).astype(np.int64)
    codes = (ips.synth.splitmix64(0x5EED0003, n) % np.uint64(D)).astype(np.uint32)
    col = dict_vals[codes]
    enc = torch.from_numpy(O.fle_encode(codes, 12).view(np.int64)).cuda()
    dd = capi.Dict(dict_vals.view(np.uint8), capi.T_INT64)
    for sel in (0.01, 0.10, 0.50, 1.0):
        lo = dict_vals[int((0.5 - sel / 2) * (D - 1))]
        hi = dict_vals[int((0.5 + sel / 2) * (D - 1))]
        _, op_lo, c_lo = dd.translate(capi.OP_GE, lo)
        _, op_hi, c_hi = dd.translate(capi.OP_LE, hi)
        kind_lo, kind_hi = dd.translate(capi.OP_GE, lo)[0], dd.translate(capi.OP_LE, hi)[0]
        exp = (col >= lo) & (col <= hi)
        a = dd.pred(enc, n, 12, capi.OP_GE, lo)
        b = dd.pred(enc, n, 12, capi.OP_LE, hi)
        got = capi.bitmap_and(a.clone(), b, n)
        got_bits = np.unpackbits(got.cpu().numpy().view(np.uint8), bitorder="little")[:n].astype(bool)
        assert np.array_equal(got_bits, exp), sel
        if kind_lo == capi.XL_FLE and kind_hi == capi.XL_FLE:
            nodes = [capi.leaf(0, op_lo, c_lo), capi.leaf(0, op_hi, c_hi), capi.and_node()]
            fused = capi.eval_program(nodes, [capi.fle_column(enc, 12)], n)
            assert torch.equal(fused, got)
        assert abs(exp.mean() - sel) < 0.02
    dd.close()


def test_dictionary_int32_in_list_and_gather(capi, ips, O):
    """configs[3]: dictionary int32, D in {256, 4096, 40000}, IN lists of K in {4, 16} literals,
    half present / half absent, plus the gather of the selected rows."""
    n = 1 << 21
    rng = np.random.default_rng(4)
    for D in (256, 4096, 40000):
        bw = capi.dict_bit_width(D)
        dict_vals = np.sort(rng.choice(np.arange(-2 ** 30, 2 ** 30, 7), D, replace=False)).astype(np.int32)
        codes = (ips.synth.splitmix64(ips.synth.SEED_DICT, n) % np.uint64(D)).astype(np.uint32)
        col = dict_vals[codes]
        enc = torch.from_numpy(O.fle_encode(codes, bw).view(np.int64)).cuda()
        dd = capi.Dict(dict_vals.view(np.uint8), capi.T_INT32)
        for K in (4, 16):
            present = dict_vals[rng.choice(D, K // 2, replace=False)]
            absent = present + 1          # step 7 grid: +1 is never in the dictionary
            lits = np.concatenate([present, absent]).astype(np.int32)
            bitmap, bvals, counts = dd.scan(enc, n, bw, capi.OP_IN, lits)
            exp = np.isin(col, present)
            got = np.unpackbits(bitmap.cpu().numpy().view(np.uint8), bitorder="little")[:n].astype(bool)
            assert np.array_equal(got, exp), (D, K)
            dense = capi.batches_compact(bvals, counts, n).cpu().numpy()
            assert np.array_equal(dense, col[exp]), (D, K)
        out, bad = dd.decode(enc, n, bw)
        assert int(bad.item()) == 0 and np.array_equal(out.cpu().numpy(), col)
        dd.close()


def test_more_than_2_31_rows(capi, ips):
    """Row indices beyond 32 bits: 2^31 + a ragged tail at w = 2 (encode, predicate, full decode,
    and the fused scan's bitmap / counts), checked against torch on the raw values."""
    n, bw = (1 << 31) + 3 * 4097 + 5, 2
    vals = capi.synth_u32(0x5EED0BB, n, bw)
    enc = capi.fle_encode(vals, bw)
    assert enc.numel() == ((n + 63) // 64) * bw
    bm = capi.fle_pred(enc, n, bw, capi.OP_GE, 2)
    exp_count = int((vals >= 2).sum().item())
    assert capi.bitmap_count(bm, n) == exp_count
    # the ragged tail: the last word has exactly (n % 64) meaningful bits and zero padding
    tail_bits = n % 64
    last = int(bm[-1].item()) & 0xFFFFFFFFFFFFFFFF
    assert last >> tail_bits == 0
    tail_vals = vals[n - tail_bits:].cpu().numpy()
    assert last == sum(1 << i for i, v in enumerate(tail_vals) if v >= 2)
    dec = capi.fle_decode(enc, n, bw, 1)
    assert torch.equal(dec.to(torch.int32)[-100000:], vals[-100000:])
    assert int(dec.sum(dtype=torch.int64).item()) == int(vals.sum(dtype=torch.int64).item())
    del dec
    # rows on both sides of the 2^31 boundary
    lo = (1 << 31) - 64
    bits = torch.tensor([(int(bm[(lo + i) // 64].item()) >> ((lo + i) % 64)) & 1 for i in range(128)])
    assert torch.equal(bits.bool(), (vals[lo:lo + 128] >= 2).cpu())


def test_optional_column_beyond_2_31_rows(capi, ips):
    """An OPTIONAL column of 2^31 + 70 001 rows (half of them NULL, w = 3): the one-pass nullable
    leaf and the late materialisation of its selection -- ranks, data-row indices and dense
    output positions beyond 32 bits -- against torch on the raw values."""
    n, bw = (1 << 31) + 70001, 3
    is_set = (capi.synth_u32(0x5EED0E1, n, 1) == 1)
    defs = capi.fle_encode(is_set.to(torch.int32), 1)
    k = int(is_set.sum().item())
    vals = capi.synth_u32(0x5EED0E2, k, bw)
    enc = capi.fle_encode(vals, bw)
    bm = capi.fle_pred_nullable(defs, 1, 1, n, enc, k, bw, capi.OP_LT, 2)
    truth = torch.zeros(n, dtype=torch.bool, device="cuda")
    pos = 0
    for c0 in range(0, n, 1 << 30):    # (boolean-mask assignment in pieces torch can index)
        piece = is_set[c0:c0 + (1 << 30)]
        cnt = int(piece.sum().item())
        truth[c0:c0 + (1 << 30)][piece] = vals[pos:pos + cnt] < 2
        pos += cnt
    n_true = int(truth.sum().item())
    assert capi.bitmap_count(bm, n) == n_true
    for lo in ((1 << 31) - 128, ((n - 300) // 64) * 64):   # both sides of 2^31, and the ragged end
        hi = min(lo + 256, n)
        words = bm[lo // 64:(hi + 63) // 64].cpu().tolist()
        bits = torch.tensor([(words[(lo + i) // 64 - lo // 64] >> ((lo + i) % 64)) & 1 for i in range(hi - lo)])
        assert torch.equal(bits.bool(), truth[lo:hi].cpu())
    assert (int(bm[-1].item()) & 0xFFFFFFFFFFFFFFFF) >> (n % 64) == 0
    del truth
    dense, flags, n_sel, n_val = capi.select_nullable(None, defs, 1, 1, n, enc, k, bw, bm)
    assert n_sel == n_val == n_true                      # every selected row is NOT NULL
    assert torch.equal(dense, vals[vals < 2])
    assert capi.bitmap_count(flags, n_sel) == n_sel


def _pack(mask):
    w = mask.view(-1, 64).to(torch.int64)
    return (w << torch.arange(64, device=w.device, dtype=torch.int64)).sum(dim=1)


def test_config2_2p28_rows_int64_between(capi, ips):
    """configs[2] at full size (2^28 rows): PLAIN int64 and dictionary int64 (D = 4096, w = 12),
    BETWEEN = And(Ge, Le) at 1 / 10 / 50 / 100 %: every bitmap bit against torch on the raw
    values, the fused PLAIN scan's selected slots against masked_select."""
    n = 1 << 28
    dev = torch.device("cuda")
    lo32 = capi.synth_u32(0x5EED0003, n, 32).to(torch.int64) & 0xFFFFFFFF
    plain = (capi.synth_u32(0x5EED1003, n, 8).to(torch.int64) << 32) | lo32      # values mod 2^40
    del lo32
    cols = [capi.plain_column(plain, capi.T_INT64)]
    for sel in (0.01, 0.10, 0.50, 1.0):
        lo = int((0.5 - sel / 2) * (1 << 40))
        hi = int((0.5 + sel / 2) * (1 << 40)) - (0 if sel < 1.0 else 1)
        nodes = [capi.plain_leaf(0, capi.OP_GE, np.int64(lo), capi.T_INT64),
                 capi.plain_leaf(0, capi.OP_LE, np.int64(hi), capi.T_INT64), capi.and_node()]
        bm = capi.eval_program(nodes, cols, n)
        mask = (plain >= lo) & (plain <= hi)
        assert torch.equal(_pack(mask), bm), sel
        assert abs(float(mask.sum().item()) / n - sel) < 0.002
        if sel == 0.10:
            bm2, bv, cnt = capi.plain_scan(plain, n, capi.T_INT64, capi.OP_GE, np.int64(lo),
                                           op2=capi.OP_LE, literal2=np.int64(hi))
            assert torch.equal(bm2, bm)
            assert torch.equal(capi.batches_compact(bv, cnt, n), torch.masked_select(plain, mask))
            bv2, cnt2 = capi.plain_select(plain, n, capi.T_INT64, bm)     # streamed batches
            assert torch.equal(cnt2, cnt)
            assert torch.equal(capi.batches_compact(bv2, cnt2, n), torch.masked_select(plain, mask))
            del bv, bv2
        del mask, bm
    del plain, cols
    torch.cuda.empty_cache()
    D = 4096
    rng = np.random.default_rng(3)
    dict_vals = np.sort(rng.choice(np.arange(-2 ** 40, 2 ** 40, 2 ** 18), D, replace=False)).astype(np.int64)
    codes = ((capi.synth_u32(0x5EED0003, n, 32).to(torch.int64) & 0xFFFFFFFF) % D).to(torch.int32)
    enc = capi.fle_encode(codes, 12)
    dd = capi.Dict(dict_vals.view(np.uint8), capi.T_INT64)
    d_dict = torch.from_numpy(dict_vals).to(dev)
    for sel in (0.01, 0.10, 0.50, 1.0):
        lo = dict_vals[int((0.5 - sel / 2) * (D - 1))]
        hi = dict_vals[int((0.5 + sel / 2) * (D - 1))]
        a = dd.pred(enc, n, 12, capi.OP_GE, lo)
        b = dd.pred(enc, n, 12, capi.OP_LE, hi)
        got = capi.bitmap_and(a.clone(), b, n)
        lo_c, hi_c = int(np.searchsorted(dict_vals, lo)), int(np.searchsorted(dict_vals, hi, side="right"))
        mask = (codes >= lo_c) & (codes < hi_c)
        assert torch.equal(_pack(mask), got), sel
        if sel == 0.10:   # the gather of the selected rows' dictionary values
            bv, cnt = dd.select(enc, n, 12, got)
            assert torch.equal(capi.batches_compact(bv, cnt, n), d_dict[torch.masked_select(codes, mask).to(torch.int64)])
            del bv
        del a, b, got, mask
    dd.close()


def test_config3_2p28_rows_dictionary_in_list(capi, ips):
    """configs[3] at full size (2^28 rows): dictionary int32, D in {256, 4096, 40000}, IN lists of
    4 and 16 literals (half absent), fused scan + gather: bitmap bits and gathered values against
    torch on the raw codes."""
    n = 1 << 28
    dev = torch.device("cuda")
    rng = np.random.default_rng(4)
    for D in (256, 4096, 40000):
        bw = capi.dict_bit_width(D)
        dict_vals = np.sort(rng.choice(np.arange(-2 ** 30, 2 ** 30, 7), D, replace=False)).astype(np.int32)
        codes = ((capi.synth_u32(ips.synth.SEED_DICT, n, 32).to(torch.int64) & 0xFFFFFFFF) % D).to(torch.int32)
        enc = capi.fle_encode(codes, bw)
        dd = capi.Dict(dict_vals.view(np.uint8), capi.T_INT32)
        d_dict = torch.from_numpy(dict_vals).to(dev)
        for K in (4, 16):
            present = rng.choice(D, K // 2, replace=False)
            lits = np.concatenate([dict_vals[present], dict_vals[present] + 1]).astype(np.int32)
            bitmap, bvals, counts = dd.scan(enc, n, bw, capi.OP_IN, lits)
            mask = torch.isin(codes, torch.tensor(np.sort(present), device=dev, dtype=torch.int32))
            assert torch.equal(_pack(mask), bitmap), (D, K)
            assert torch.equal(capi.batches_compact(bvals, counts, n),
                               d_dict[torch.masked_select(codes, mask).to(torch.int64)]), (D, K)
            del bitmap, bvals, counts, mask
        dd.close()
        del codes, enc
        torch.cuda.empty_cache()


# ---- configs[0]/[1], the PLAIN twin (BASELINE.md 4 "1/2 PLAIN twin"): the same
# splitmix64(0x5EED0001 + i) column as little-endian int32 slots (reinterpreted signed), LT with the
# reference's reversed operands (parquet-common.h:208-217: bit = literal < x) and with SQL's ----
def _bits(words_, n):
    return np.unpackbits(np.ascontiguousarray(words_).view(np.uint8), bitorder="little")[:n].astype(bool)


def test_config1_plain_int32_twin_1m_rows_bit_exact(capi, ips, O):
    n = 1 << 20
    vals = ips.synth.column_u32(ips.synth.SEED_HEADLINE, n, 32).view(np.int32)
    c = np.int32(ips.synth.lt_constant(32))
    page = O.plain_encode(vals, O.T_INT32)
    d_page = capi.synth_u32(ips.synth.SEED_HEADLINE, n, 32)          # the generator twin, int32 slots
    assert np.array_equal(d_page.cpu().numpy().view(np.uint8), page)
    for sem, truth in ((O.SEM_REFERENCE, c < vals), (O.SEM_SQL, vals < c)):
        ref = O.plain_pred(page, n, O.T_INT32, O.OP_LT, c, sem)      # parquet-common.h:208-217 / SQL order
        assert np.array_equal(_bits(ref, n), truth), sem
        assert np.array_equal(words(capi.plain_pred(d_page, n, capi.T_INT32, capi.OP_LT, c, sem)), ref), sem
        bm, bv, cnt = capi.plain_scan(d_page, n, capi.T_INT32, capi.OP_LT, c, semantics=sem)
        assert np.array_equal(words(bm), ref), sem
        dense = capi.batches_compact(bv, cnt, n).cpu().numpy()
        assert np.array_equal(dense, vals[truth]), sem                # ReadValue(skip) of the selected rows
        bv2, cnt2 = capi.plain_select(d_page, n, capi.T_INT32, bm)
        assert torch.equal(cnt2, cnt)
        assert np.array_equal(capi.batches_compact(bv2, cnt2, n).cpu().numpy(), vals[truth]), sem
    assert abs((vals < c).mean() - 0.60) < 0.005 and abs((c < vals).mean() - 0.40) < 0.005


def test_config2_plain_int32_twin_2p28_rows(capi, ips):
    """The PLAIN int32 twin at 2^28 rows: every bitmap bit and every selected slot against torch on
    the raw values, both operand orders; predicate, fused scan and select."""
    n = 1 << 28
    c = int(ips.synth.lt_constant(32))
    page = capi.synth_u32(ips.synth.SEED_HEADLINE, n, 32)            # int32 slots
    for sem in (capi.SEM_REFERENCE, capi.SEM_SQL):
        mask = (page > c) if sem == capi.SEM_REFERENCE else (page < c)
        packed = _pack(mask)
        assert torch.equal(capi.plain_pred(page, n, capi.T_INT32, capi.OP_LT, np.int32(c), sem), packed), sem
        bm, bv, cnt = capi.plain_scan(page, n, capi.T_INT32, capi.OP_LT, np.int32(c), semantics=sem)
        assert torch.equal(bm, packed), sem
        exp = torch.masked_select(page, mask)
        assert int(cnt.to(torch.int64).sum().item()) == exp.numel()
        assert torch.equal(capi.batches_compact(bv, cnt, n), exp), sem
        del bv
        bv2, cnt2 = capi.plain_select(page, n, capi.T_INT32, bm)
        assert torch.equal(cnt2, cnt)
        assert torch.equal(capi.batches_compact(bv2, cnt2, n), exp), sem
        del bv2, exp, mask, packed, bm
        torch.cuda.empty_cache()


def test_config2_2p28_rows_int64_and_gt_lt(capi, ips):
    """configs[2]'s second form, And(Gt a, Lt b) (simple-predicates.h:145-153), at 2^28
    rows on the PLAIN int64 column and on the D = 4096 dictionary, 1 / 10 / 50 / 100 %: every bitmap
    bit against torch on the raw values; at 10 % also the selected values."""
    n = 1 << 28
    dev = torch.device("cuda")
    lo32 = capi.synth_u32(0x5EED0003, n, 32).to(torch.int64) & 0xFFFFFFFF
    plain = (capi.synth_u32(0x5EED1003, n, 8).to(torch.int64) << 32) | lo32      # values mod 2^40
    del lo32
    cols = [capi.plain_column(plain, capi.T_INT64)]
    for sel in (0.01, 0.10, 0.50, 1.0):
        a = int((0.5 - sel / 2) * (1 << 40)) - 1           # x > a AND x < b: the open interval
        b = int((0.5 + sel / 2) * (1 << 40))
        nodes = [capi.plain_leaf(0, capi.OP_GT, np.int64(a), capi.T_INT64),
                 capi.plain_leaf(0, capi.OP_LT, np.int64(b), capi.T_INT64), capi.and_node()]
        bm = capi.eval_program(nodes, cols, n)
        mask = (plain > a) & (plain < b)
        assert torch.equal(_pack(mask), bm), sel
        assert abs(float(mask.sum().item()) / n - sel) < 0.002
        # the two leaves one by one (ParquetPlainEncoder::Gt / Lt, SQL order) and their AND
        g = capi.plain_pred(plain, n, capi.T_INT64, capi.OP_GT, np.int64(a))
        l = capi.plain_pred(plain, n, capi.T_INT64, capi.OP_LT, np.int64(b))
        assert torch.equal(capi.bitmap_and(g, l, n), bm), sel
        if sel == 0.10:
            bm2, bv, cnt = capi.plain_scan(plain, n, capi.T_INT64, capi.OP_GT, np.int64(a),
                                           op2=capi.OP_LT, literal2=np.int64(b))
            assert torch.equal(bm2, bm)
            assert torch.equal(capi.batches_compact(bv, cnt, n), torch.masked_select(plain, mask))
            del bv
        del mask, bm, g, l
    del plain, cols
    torch.cuda.empty_cache()
    D = 4096
    rng = np.random.default_rng(3)
    dict_vals = np.sort(rng.choice(np.arange(-2 ** 40, 2 ** 40, 2 ** 18), D, replace=False)).astype(np.int64)
    codes = ((capi.synth_u32(0x5EED0003, n, 32).to(torch.int64) & 0xFFFFFFFF) % D).to(torch.int32)
    enc = capi.fle_encode(codes, 12)
    dd = capi.Dict(dict_vals.view(np.uint8), capi.T_INT64)
    d_dict = torch.from_numpy(dict_vals).to(dev)
    for sel in (0.01, 0.10, 0.50, 1.0):
        ia, ib = int((0.5 - sel / 2) * (D - 1)), int((0.5 + sel / 2) * (D - 1))
        a = dict_vals[ia] - (1 if sel == 1.0 else 0)       # 100 %: a below the first entry, b above the last
        b = dict_vals[ib] + (1 if sel == 1.0 else 0)
        mask = (codes > (ia if sel < 1.0 else -1)) & (codes < (ib if sel < 1.0 else D))
        packed = _pack(mask)
        g = dd.pred(enc, n, 12, capi.OP_GT, a)             # DictDecoder::Gt -> Ge(upper_bound), dict-encoding.h:473-483
        l = dd.pred(enc, n, 12, capi.OP_LT, b)             # DictDecoder::Lt -> Lt(lower_bound), :485-495
        assert torch.equal(capi.bitmap_and(g.clone(), l, n), packed), sel
        ka, op_a, c_a = dd.translate(capi.OP_GT, a)
        kb, op_b, c_b = dd.translate(capi.OP_LT, b)
        if ka == capi.XL_FLE and kb == capi.XL_FLE:         # both leaves on the codes, one pass
            nodes = [capi.leaf(0, op_a, c_a), capi.leaf(0, op_b, c_b), capi.and_node()]
            assert torch.equal(capi.eval_program(nodes, [capi.fle_column(enc, 12)], n), packed), sel
        else:
            assert sel == 1.0
        if sel == 0.10:
            bv, cnt = dd.select(enc, n, 12, packed)
            assert torch.equal(capi.batches_compact(bv, cnt, n), d_dict[torch.masked_select(codes, mask).to(torch.int64)])
            del bv
        del g, l, mask, packed
    dd.close()


def test_page_lists_at_full_size(capi, ips):
    """configs[4] and configs[3] over SEPARATE page buffers (ips_chunk_*): the Q6 conjunction over
    600,037,902 rows with the three columns cut into pages of different sizes (2^20, 2^20 - 37 and
    700,001 rows: every page boundary of every column falls inside a bitmap word), and the
    dictionary IN scan + gather over 257 unaligned pages -- the very words / values of the
    contiguous-buffer calls (themselves checked against torch above)."""
    q6 = ips.q6
    n = q6.ROWS
    codes = [q6.codes_gpu(capi, c, n) for c in range(3)]
    encs = [capi.fle_encode(codes[c], q6.COLUMNS[c][3]) for c in range(3)]
    nodes, cols = q6.program(capi, encs)
    ref = capi.eval_program(nodes, cols, n)
    assert torch.equal(ref, _pack_rows(q6.truth(codes)))
    del encs, cols

    def chunk_of(vals, w, rows):
        pages, pos = [], 0
        while pos < vals.numel():
            m = min(rows, vals.numel() - pos)
            pages.append((capi.fle_encode(vals[pos:pos + m].clone(), w), m, w))
            pos += m
        return capi.Chunk(pages)
    chunks = [chunk_of(codes[c], q6.COLUMNS[c][3], r) for c, r in enumerate((1 << 20, (1 << 20) - 37, 700001))]
    got = capi.eval_program_chunks(nodes, chunks)
    assert torch.equal(got, ref)
    for ch in chunks:
        ch.close()
    del chunks, codes, got, ref
    torch.cuda.empty_cache()
    n = 1 << 28
    rng = np.random.default_rng(4)
    D, bw = 4096, 12
    dict_vals = np.sort(rng.choice(np.arange(-2 ** 30, 2 ** 30, 7), D, replace=False)).astype(np.int32)
    cds = ((capi.synth_u32(ips.synth.SEED_DICT, n, 32).to(torch.int64) & 0xFFFFFFFF) % D).to(torch.int32)
    dd = capi.Dict(dict_vals.view(np.uint8), capi.T_INT32)
    present = rng.choice(D, 8, replace=False)
    lits = np.concatenate([dict_vals[present], dict_vals[present] + 1]).astype(np.int32)
    chunk = chunk_of(cds, bw, (1 << 20) - 37)
    bitmap, bvals, counts = chunk.dict_scan(dd, capi.OP_IN, lits)
    mask = torch.isin(cds, torch.tensor(np.sort(present), device="cuda", dtype=torch.int32))
    assert torch.equal(bitmap, _pack_rows(mask))
    d_dict = torch.from_numpy(dict_vals).cuda()
    assert torch.equal(chunk.compact(bvals, counts), d_dict[torch.masked_select(cds, mask).to(torch.int64)])
    chunk.close()
    dd.close()


def _pack_rows(mask):
    n = mask.numel()
    pad = (-n) % 64
    if pad:
        mask = torch.cat([mask, torch.zeros(pad, dtype=torch.bool, device=mask.device)])
    return _pack(mask)


def test_one_pass_chain_with_more_stripes_than_waves(capi, request):
    """The one-pass conjunct chain launches one 2048-row stripe per wave up to 64 workgroups per resident
    slot; beyond that (here 2^31 + 70 000 rows of two 1-bit columns: 1 048 611 stripes) a wave loops over several
    stripes with the next one prefetched.  A 1-bit FLE column IS its plane words (row k of a block at bit
    63 - k), so random words are the encoded columns and (c0 == 1) AND (c1 == 1) must be the bit-reversed
    AND of the words: popcount over everything, every word of the first and last 2^16 blocks, and equality
    with the per-operand plan."""
    request.addfinalizer(lambda: capi.set_program_strategy(capi.PROGRAM_AUTO))
    n = (1 << 31) + 70000
    n_words = (n + 63) // 64
    g = torch.Generator(device="cuda")
    g.manual_seed(2031)
    enc = [torch.randint(-(1 << 63), (1 << 63) - 1, (n_words + 2,), dtype=torch.int64, device="cuda", generator=g)
           for _ in range(2)]
    cols = [capi.fle_column(enc[0], 1), capi.fle_column(enc[1], 1)]
    nodes = [capi.leaf(0, capi.OP_EQ, 1), capi.leaf(1, capi.OP_EQ, 1), capi.and_node()]
    capi.set_program_strategy(capi.PROGRAM_ONE_PASS)
    got = capi.eval_program(nodes, cols, n).clone()
    both = enc[0][:n_words] & enc[1][:n_words]
    tail = n % 64  # rows of the last block: its high 'tail' bits
    both[-1] &= -(1 << (64 - tail))
    assert capi.bitmap_count(got, n) == capi.bitmap_count(both, n_words * 64)

    def bitrev(a):
        a = a.view(np.uint8).reshape(-1, 8)[:, ::-1]
        return np.packbits(np.unpackbits(a, axis=1, bitorder="big"), axis=1, bitorder="little").view(np.uint64).ravel()

    for lo, hi in ((0, 1 << 16), (n_words - (1 << 16), n_words), (n_words // 2, n_words // 2 + 4096)):
        assert np.array_equal(words(got[lo:hi]), bitrev(both[lo:hi].cpu().numpy())), (lo, hi)
    capi.set_program_strategy(capi.PROGRAM_PER_OPERAND)
    ref = capi.eval_program(nodes, cols, n)
    assert torch.equal(ref, got)

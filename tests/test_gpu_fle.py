"""GPU parity tests (through the C-ABI) of the FLE kernels against the oracle: encode, decode,
predicate-on-encoded, fused scan, late materialisation.  Bit-exact (integer work)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SIZES = [1, 63, 64, 65, 100, 2047, 2048, 2049, 4096 + 17, 20011]


def enc_to_dev(enc):
    """Encoded words at their exact size (the kernels must not read past ceil(n/64)*w words)."""
    return torch.from_numpy(np.ascontiguousarray(enc).view(np.int64)).cuda() if len(enc) else \
        torch.zeros(2, dtype=torch.int64, device="cuda")


def words(t):
    return t.cpu().numpy().view(np.uint64)


def rand_vals(rng, n, bw):
    return rng.integers(0, 1 << bw, n, dtype=np.uint64).astype(np.uint32)


@pytest.mark.parametrize("bw", range(1, 33))
def test_encode_decode_vs_oracle(capi, O, bw):
    rng = np.random.default_rng(100 + bw)
    for n in SIZES:
        vals = rand_vals(rng, n, bw)
        ref = O.fle_encode(vals, bw)
        d_vals = torch.from_numpy(np.concatenate([vals, np.zeros(8, np.uint32)]).view(np.int32)).cuda()
        enc = capi.fle_encode(d_vals[:n], bw)
        torch.cuda.synchronize()
        assert np.array_equal(words(enc), ref), (bw, n)
        out = capi.fle_decode(enc_to_dev(ref), n, bw, 4)
        assert np.array_equal(out.cpu().numpy().view(np.uint32), vals), (bw, n)


@pytest.mark.parametrize("bw", [1, 3, 8, 9, 13, 16])
def test_narrow_widths(capi, O, bw):
    """Reference staging widths: u8 for w<=8, u16 for w<=16 (fle-encoding.h:365-371)."""
    rng = np.random.default_rng(bw)
    for n in (65, 2048 + 5, 10007):
        vals = rand_vals(rng, n, bw)
        ref = O.fle_encode(vals, bw)
        for width, npt, tt in ((1, np.uint8, torch.uint8), (2, np.uint16, torch.int16)):
            if bw > 8 * width:
                continue
            out = capi.fle_decode(enc_to_dev(ref), n, bw, width)
            assert np.array_equal(out.cpu().numpy().view(npt), vals.astype(npt))
            src = torch.from_numpy(np.concatenate([vals.astype(npt), np.zeros(16, npt)])
                                   .view(np.int16 if width == 2 else np.uint8)).cuda()
            enc = capi.fle_encode(src[:n], bw)
            assert np.array_equal(words(enc), ref)


@pytest.mark.parametrize("bw", range(1, 33))
def test_pred_vs_oracle(capi, O, bw):
    rng = np.random.default_rng(200 + bw)
    for n in SIZES:
        vals = rand_vals(rng, n, bw)
        ref_enc = O.fle_encode(vals, bw)
        enc = enc_to_dev(ref_enc)
        consts = sorted({0, (1 << bw) - 1, int(vals[0]), (1 << bw) // 10})
        for c in consts:
            for op in (O.OP_EQ, O.OP_LT, O.OP_LE, O.OP_GT, O.OP_GE):
                got = words(capi.fle_pred(enc, n, bw, op, c))
                assert np.array_equal(got, O.fle_pred(ref_enc, n, bw, op, c)), (bw, n, op, c)
        lst = sorted({int(vals[0]), int(vals[-1]), 0, (1 << bw) - 1})
        got = words(capi.fle_pred(enc, n, bw, O.OP_IN, lst))
        assert np.array_equal(got, O.fle_pred(ref_enc, n, bw, O.OP_IN, lst)), (bw, n)


@pytest.mark.parametrize("bw", [3, 8, 12, 16, 17, 32])
def test_in_list_lengths(capi, O, bw):
    """IN lists of every length class of the kernels' constant loops (rounds of four + remainder,
    up to IPS_MAX_IN_LIST = 256), predicate-only and fused scan; duplicates allowed."""
    rng = np.random.default_rng(900 + bw)
    n = 6000
    vals = rand_vals(rng, n, bw)
    ref_enc = O.fle_encode(vals, bw)
    enc = enc_to_dev(ref_enc)
    for k in (1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 63, 255, 256):
        lst = [int(x) for x in rng.choice(vals, k)]          # present values, maybe repeated
        lst[k // 2] = int(rng.integers(0, 1 << bw))           # plus one that may be absent
        ref = O.fle_pred(ref_enc, n, bw, O.OP_IN, lst)
        assert np.array_equal(words(capi.fle_pred(enc, n, bw, O.OP_IN, lst)), ref), (bw, k)
        bitmap, bvals, counts = capi.fle_scan(enc, n, bw, O.OP_IN, lst)
        assert np.array_equal(words(bitmap), ref), (bw, k)
        check_batches(bvals, counts, n, O.fle_select(ref_enc, n, bw, ref), capi)


@pytest.mark.parametrize("bw", [9, 12, 16])
def test_in_list_scan_dense_selection(capi, O, bw):
    """A short IN list that selects most rows of a w = 9..16 column (skewed values): the fused scan
    works its index list off in several windows of 512 entries per sub-tile."""
    rng = np.random.default_rng(950 + bw)
    for n in (2048, 2049, 30011):
        hot = rng.choice(1 << bw, 3, replace=False)
        vals = np.where(rng.random(n) < 0.7, hot[rng.integers(0, 3, n)], rng.integers(0, 1 << bw, n)).astype(np.uint32)
        ref_enc = O.fle_encode(vals, bw)
        enc = enc_to_dev(ref_enc)
        lst = [int(x) for x in hot] + [int(rng.integers(0, 1 << bw)), int(vals[0])]
        ref = O.fle_pred(ref_enc, n, bw, O.OP_IN, lst)
        bitmap, bvals, counts = capi.fle_scan(enc, n, bw, O.OP_IN, lst)
        assert np.array_equal(words(bitmap), ref), (bw, n)
        assert counts.cpu().numpy().max() > 1200 or n < 2049
        check_batches(bvals, counts, n, O.fle_select(ref_enc, n, bw, ref), capi)


def check_batches(bvals, counts, n, expect_dense, capi):
    counts_h = counts.cpu().numpy()
    b_h = bvals.cpu().numpy().view(np.uint32)
    got = np.concatenate([b_h[b * 2048: b * 2048 + counts_h[b]] for b in range(len(counts_h))]) \
        if len(counts_h) else np.zeros(0, np.uint32)
    assert np.array_equal(got, expect_dense)
    dense = capi.batches_compact(bvals, counts, n).cpu().numpy().view(np.uint32)
    assert np.array_equal(dense, expect_dense)


@pytest.mark.parametrize("bw", range(1, 33))
def test_fused_scan_vs_oracle(capi, O, bw):
    rng = np.random.default_rng(300 + bw)
    for n in (1, 64, 2047, 2049, 20011):
        vals = rand_vals(rng, n, bw)
        ref_enc = O.fle_encode(vals, bw)
        enc = enc_to_dev(ref_enc)
        cases = [(O.OP_LT, (1 << bw) // 10 + 1), (O.OP_GE, (1 << bw) // 2), (O.OP_EQ, int(vals[0])),
                 (O.OP_LE, (1 << bw) - 1), (O.OP_GT, (1 << bw) - 1),
                 (O.OP_IN, sorted({int(vals[0]), int(vals[n // 2]), 0}))]
        for op, c in cases:
            bitmap, bvals, counts = capi.fle_scan(enc, n, bw, op, c)
            bm_ref = O.fle_pred(ref_enc, n, bw, op, c)
            assert np.array_equal(words(bitmap), bm_ref), (bw, n, op)
            # late materialisation == the reference's Get(val, skip) walk
            check_batches(bvals, counts, n, O.fle_select(ref_enc, n, bw, bm_ref), capi)


@pytest.mark.parametrize("bw", [1, 7, 12, 20, 32])
def test_select_given_bitmap(capi, O, bw):
    rng = np.random.default_rng(400 + bw)
    for n in (65, 4096, 10000):
        vals = rand_vals(rng, n, bw)
        ref_enc = O.fle_encode(vals, bw)
        for p in (0.0, 0.02, 0.5, 1.0):
            bits = rng.random(n) < p
            bm = np.packbits(bits, bitorder="little")
            bm = np.concatenate([bm, np.zeros(-len(bm) % 8, np.uint8)]).view(np.uint64)
            bvals, counts = capi.fle_select(enc_to_dev(ref_enc), n, bw, enc_to_dev(bm))
            check_batches(bvals, counts, n, vals[bits], capi)


def test_out_of_range_constants(capi, O):
    """c >= 2^w: defined by the unsigned SQL meaning, never UB."""
    bw, n = 5, 1000
    vals = (np.arange(n) % 32).astype(np.uint32)
    enc = enc_to_dev(O.fle_encode(vals, bw))
    full = np.ones(n, bool)
    for op, expect in ((O.OP_LT, full), (O.OP_LE, full), (O.OP_GT, ~full), (O.OP_GE, ~full),
                       (O.OP_EQ, ~full)):
        got = np.unpackbits(words(capi.fle_pred(enc, n, bw, op, 32)).view(np.uint8),
                            bitorder="little")[:n].astype(bool)
        assert np.array_equal(got, expect)
    got = words(capi.fle_pred(enc, n, bw, O.OP_IN, [3, 77]))
    assert np.array_equal(got, O.fle_pred(O.fle_encode(vals, bw), n, bw, O.OP_IN, [3]))


def test_known_answer_on_gpu(capi, O):
    """i % 32 at w=5, Lt(7) over rows 0..63 -> 0x0000007f0000007f."""
    vals = (np.arange(64) % 32).astype(np.uint32)
    d_vals = torch.from_numpy(vals.view(np.int32)).cuda()
    enc = capi.fle_encode(d_vals, 5)
    assert [hex(x) for x in words(enc)] == ["0x5555555555555555", "0x3333333333333333",
                                             "0xf0f0f0f0f0f0f0f", "0xff00ff00ff00ff",
                                             "0xffff0000ffff"]
    assert int(words(capi.fle_pred(enc, 64, 5, O.OP_LT, 7))[0]) == 0x0000007F0000007F


def test_reference_padding_is_ignored(capi, O):
    """The reference encoder leaves garbage in the padding rows of the last block (quirk Q4):
    bitmaps and decoded rows must not depend on it."""
    bw, n = 9, 100
    rng = np.random.default_rng(1)
    vals = rand_vals(rng, n, bw)
    clean = O.fle_encode(vals, bw)
    dirty = O.fle_encode(np.concatenate([vals, rand_vals(rng, 28, bw)]), bw)
    assert not np.array_equal(clean, dirty)
    for op in range(5):
        a = words(capi.fle_pred(enc_to_dev(clean), n, bw, op, 100))
        b = words(capi.fle_pred(enc_to_dev(dirty), n, bw, op, 100))
        assert np.array_equal(a, b)
    assert np.array_equal(capi.fle_decode(enc_to_dev(dirty), n, bw).cpu().numpy().view(np.uint32), vals)


@pytest.mark.parametrize("bw", [1, 7, 12, 16, 21, 32])
def test_scan_pages_equals_per_page_scan(capi, O, bw):
    """ips_fle_scan_pages over separate page buffers of ragged sizes (empty, one row, below and
    above one batch, above the 64-page launch group) equals ips_fle_scan page by page and the
    oracle."""
    rng = np.random.default_rng(1200 + bw)
    sizes = [0, 1, 63, 2048, 2049, 5000, 70001] + [int(x) for x in rng.integers(1, 9000, 70)]
    pages, refs = [], []
    for n in sizes:
        vals = rand_vals(rng, max(n, 1), bw)[:n]
        ref_enc = O.fle_encode(vals, bw) if n else np.zeros(0, np.uint64)
        enc = enc_to_dev(ref_enc) if n else torch.zeros(2, dtype=torch.int64, device="cuda")
        outs = capi.alloc_scan_outputs(n, torch.device("cuda"))
        pages.append((enc, n, outs))
        refs.append((ref_enc, vals))
    plist = capi.make_page_list(pages)
    for op, c in ((O.OP_LT, (1 << bw) // 10 + 1), (O.OP_GE, (1 << bw) // 2), (O.OP_EQ, int(refs[5][1][0])),
                  (O.OP_LE, (1 << bw) - 1), (O.OP_GT, (1 << bw) - 1)):
        for _, _, outs in pages:
            for t in outs:
                t.fill_(-1)
        capi.fle_scan_pages(plist, bw, op, c)
        for (enc, n, outs), (ref_enc, vals) in zip(pages, refs):
            if n == 0:
                continue
            bm_ref = O.fle_pred(ref_enc, n, bw, op, c)
            assert np.array_equal(words(outs[0][:(n + 63) // 64]), bm_ref), (bw, n, op)
            check_batches(outs[1], outs[2][:(n + 2047) // 2048], n, O.fle_select(ref_enc, n, bw, bm_ref), capi)
    with pytest.raises(capi.IpsError):
        capi.fle_scan_pages(plist, bw, O.OP_IN, [0, 1])

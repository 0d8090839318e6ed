"""BASELINE configs[4] on the HIP kernels: the TPC-H-Q6-shaped three-column conjunction
(EvalSimplePredicates' conjunct AND, hdfs-parquet-scanner.cc:1857-1862) at the
full 600,037,902 rows on one GPU, and row stripes cut by sharding.stripe_bounds, scanned one by
one by the HIP kernels and concatenated, against the unsharded result."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def words(t):
    return t.cpu().numpy().view(np.uint64)


def oracle_q6_bitmap(O, q6, n, start=0):
    """The conjunction leaf by leaf through the oracle's FleDecoder predicates + AND."""
    ops = {"GE": O.OP_GE, "LT": O.OP_LT}
    encs = [O.fle_encode(q6.codes_numpy(c, n, start), q6.COLUMNS[c][3]) for c in range(3)]
    bm = None
    for col, op, k in q6.LEAVES:
        leaf = O.fle_pred(encs[col], n, q6.COLUMNS[col][3], ops[op], k)
        bm = leaf if bm is None else (bm & leaf)
    return bm


def pack_mask(mask):
    """bool tensor -> LSB-first int64 bitmap words (zero padded), on the GPU."""
    n = mask.numel()
    pad = (-n) % 64
    if pad:
        mask = torch.cat([mask, torch.zeros(pad, dtype=torch.bool, device=mask.device)])
    w = mask.view(-1, 64).to(torch.int64)
    return (w << torch.arange(64, device=w.device, dtype=torch.int64)).sum(dim=1)


def test_config4_q6_600m_rows_one_gpu(capi, ips, O):
    q6 = ips.q6
    n = q6.ROWS
    codes = [q6.codes_gpu(capi, c, n) for c in range(3)]
    encs = [capi.fle_encode(codes[c], q6.COLUMNS[c][3]) for c in range(3)]
    nodes, cols = q6.program(capi, encs)
    bm = capi.eval_program(nodes, cols, n)
    mask = q6.truth(codes)
    n_sel = int(mask.sum().item())
    # 365/2526 * 3/11 * 23/50 = 1.81 % of the rows
    assert abs(n_sel / n - (365 / 2526) * (3 / 11) * (23 / 50)) < 2e-4
    assert capi.bitmap_count(bm, n) == n_sel
    # every bit, against torch on the raw codes; the padding bits of the last word are zero
    exp = pack_mask(mask)
    assert torch.equal(exp, bm)
    assert (int(bm[-1].item()) & 0xFFFFFFFFFFFFFFFF) >> (n % 64) == 0
    del exp, mask
    # oracle-exact on the first 2^20 rows
    n1 = 1 << 20
    assert np.array_equal(words(bm[:n1 // 64]), oracle_q6_bitmap(O, q6, n1))
    # idempotence and the one-launch interpreter agree with the per-operand plan
    assert torch.equal(capi.eval_program(nodes, cols, n), bm)
    # the scan carried on: late materialisation of l_quantity's codes against the bitmap
    bvals, counts = capi.fle_select(encs[2], n, 6, bm)
    assert int(counts.to(torch.int64).sum().item()) == n_sel
    dense = capi.batches_compact(bvals, counts, n)
    sel_codes = torch.masked_select(codes[2], q6.truth(codes))
    assert torch.equal(dense, sel_codes)
    assert int(dense.max().item()) < 23


@pytest.mark.parametrize("n", [2048 * 37 + 777, 3000, 2048 * 8, (1 << 22) + 12345])
@pytest.mark.parametrize("world", [2, 4, 8])
def test_stripes_scanned_by_the_hip_kernels_concatenate(capi, ips, O, n, world):
    """Cut the three columns with sharding.stripe_bounds for `world` ranks, run ips_eval_program
    and the fused ips_fle_scan on every stripe with the HIP kernels, concatenate: the result is
    the unsharded bitmap / batches (ragged last stripe, empty stripes included)."""
    q6, sh = ips.q6, ips.sharding
    codes = [q6.codes_gpu(capi, c, n) for c in range(3)]
    encs = [capi.fle_encode(codes[c], q6.COLUMNS[c][3]) for c in range(3)]
    nodes, cols = q6.program(capi, encs)
    whole = capi.eval_program(nodes, cols, n)
    w_bm, w_vals, w_cnt = capi.fle_scan(encs[0], n, 12, capi.OP_LT, 365)
    assert np.array_equal(words(whole), oracle_q6_bitmap(O, q6, n))

    s_rows = sh.stripe_rows(n, world)
    assert s_rows % 2048 == 0
    pieces, scan_pieces, cnt_pieces, val_pieces = [], [], [], []
    for rank in range(world):
        row0, row1 = sh.stripe_bounds(n, world, rank)
        m = row1 - row0
        if m == 0:  # ranks beyond the data hold an empty stripe
            continue
        senc = [sh.stripe_word_slice(encs[c], q6.COLUMNS[c][3], n, world, rank) for c in range(3)]
        for c in range(3):
            assert senc[c].data_ptr() % 16 == 0
            assert senc[c].numel() == ((m + 63) // 64) * q6.COLUMNS[c][3]
        snodes, scols = q6.program(capi, senc)
        pieces.append(capi.eval_program(snodes, scols, m))
        bm, vals, cnt = capi.fle_scan(senc[0], m, 12, capi.OP_LT, 365)
        scan_pieces.append(bm)
        cnt_pieces.append(cnt)
        val_pieces.append(vals[:cnt.numel() * 2048])
        # a stripe regenerated from its own row offset holds the same codes (what rank r would do)
        assert torch.equal(q6.codes_gpu(capi, 0, min(m, 4096), start=row0), codes[0][row0:row0 + min(m, 4096)])
    # every stripe but the last is a whole number of bitmap words and of batches
    assert torch.equal(torch.cat(pieces), whole)
    assert torch.equal(torch.cat(scan_pieces), w_bm)
    assert torch.equal(torch.cat(cnt_pieces), w_cnt)
    nb = w_cnt.numel()
    assert torch.equal(capi.batches_compact(torch.cat(val_pieces), torch.cat(cnt_pieces), n),
                       capi.batches_compact(w_vals[:nb * 2048], w_cnt, n))


def test_chunked_scan_with_overlapped_allgather_one_rank(capi, ips, O):
    """ips_fle_scan_allgather (one C call per step: chunk scans + RCCL all-gather of every finished
    chunk on the communicator's stream) on a one-rank communicator: the gathered bitmap, the local
    bitmap and the batches equal a plain ips_fle_scan of the same rows and the oracle."""
    n_chunks, bw = 8, 13
    n = 2048 * n_chunks * 5
    vals = ips.synth.column_u32(ips.synth.SEED_HEADLINE, n, bw)
    c = ips.synth.lt_constant(bw)
    enc_ref = O.fle_encode(vals, bw)
    enc = torch.from_numpy(enc_ref.view(np.int64).copy()).cuda()
    comm = capi.Comm(capi.comm_unique_id(), 1, 0)
    try:
        local, bvals, counts = capi.alloc_scan_outputs(n, enc.device)
        full = torch.zeros(n // 64, dtype=torch.int64, device=enc.device)
        for _ in range(3):   # repeated steps reuse the buffers behind ips_comm_join
            comm.join()
            comm.fle_scan_allgather(enc, n, bw, capi.OP_LT, c, n_chunks, local, bvals, counts, full)
        comm.join()
        torch.cuda.synchronize()
        comm.check()         # no waiter of a piece gave up
        ref_bm, ref_vals, ref_cnt = capi.fle_scan(enc, n, bw, capi.OP_LT, c)
        assert np.array_equal(words(full), O.fle_pred(enc_ref, n, bw, O.OP_LT, c))
        assert torch.equal(full, ref_bm) and torch.equal(local[:n // 64], ref_bm)
        assert torch.equal(counts[:n // 2048], ref_cnt)
        assert torch.equal(capi.batches_compact(bvals, counts[:n // 2048], n), capi.batches_compact(ref_vals, ref_cnt, n))
        with pytest.raises(capi.IpsError):   # rows must be whole chunks of whole batches
            comm.fle_scan_allgather(enc, n - 2048, bw, capi.OP_LT, c, n_chunks, local, bvals, counts, full)
    finally:
        comm.close()


def test_sharded_program_step_one_rank(capi, ips, O):
    """ips_eval_program_chunks_allgather on a one-rank communicator: the Q6 conjunction over column
    chunks cut into 8 exchange pieces (one C call per step: the plan's launches over all pieces, the
    last operand signalling piece after piece, a waiter + all-gather per piece on the communicator's
    stream) equals ips_eval_program on the contiguous columns and the oracle; also a tree whose last
    operand cannot signal (an OR of two bitmaps: merge kernel) and one on an OPTIONAL column."""
    q6 = ips.q6
    n_pieces, piece = 8, 2048 * 9
    n = n_pieces * piece
    codes = [q6.codes_numpy(c, n) for c in range(3)]
    chunks, encs = [], []
    for c in range(3):
        w = q6.COLUMNS[c][3]
        enc = torch.from_numpy(O.fle_encode(codes[c], w).view(np.int64).copy()).cuda()
        encs.append(enc)
        wpp = piece // 64 * w
        chunks.append(capi.Chunk([(enc[i * wpp:(i + 1) * wpp].clone(), piece, w) for i in range(n_pieces)]))
    nodes, cols = q6.program(capi, encs)
    ref = capi.eval_program(nodes, cols, n)
    ops = {"GE": O.OP_GE, "LT": O.OP_LT}
    exp = None
    for col, op, k in q6.LEAVES:
        leaf = O.fle_pred(O.fle_encode(codes[col], q6.COLUMNS[col][3]), n, q6.COLUMNS[col][3], ops[op], k)
        exp = leaf if exp is None else (exp & leaf)
    assert np.array_equal(words(ref), exp)
    comm = capi.Comm(capi.comm_unique_id(), 1, 0)
    try:
        local = torch.zeros(n // 64, dtype=torch.int64, device="cuda")
        full = torch.zeros(n // 64, dtype=torch.int64, device="cuda")
        for _ in range(3):
            comm.eval_program_chunks_allgather(nodes, chunks, local, full)
        comm.join()
        torch.cuda.synchronize()
        comm.check()
        assert torch.equal(full, ref) and torch.equal(local, ref)
        # (A and B) or (C and D): the last step merges two bitmaps -- all pieces are released at once
        L, AND, OR = capi.leaf, capi.and_node, capi.or_node
        tree = [L(0, capi.OP_GE, 365), L(1, capi.OP_LT, 3), AND(), L(2, capi.OP_GE, 40), L(1, capi.OP_GE, 9), AND(), OR()]
        ws = torch.empty(n // 8 + 4096, dtype=torch.uint8, device="cuda")
        comm.eval_program_chunks_allgather(tree, chunks, local, full, workspace=ws)
        comm.join()
        torch.cuda.synchronize()
        comm.check()
        t = ((codes[0] >= 365) & (codes[1] < 3)) | ((codes[2] >= 40) & (codes[1] >= 9))
        assert np.array_equal(words(full), np.packbits(t, bitorder="little").view(np.uint64))
    finally:
        comm.close()
        for ch in chunks:
            ch.close()

"""CPU test of the multi-GPU path's host logic (world_size 2, gloo): row stripes are cut on
2048-row boundaries, each rank produces the bitmap of its stripe (the oracle stands in for the HIP
kernel here -- no GPU in this container), the stripes' bitmap words are all-gathered, and the
result must be bit-identical to the single-process bitmap of the whole column."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_rows, bw, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as g
    from oracle import oracle as O
    sharding = g.load_package().sharding
    synth = g.load_package().synth
    vals = synth.column_u32(synth.SEED_HEADLINE, n_rows, bw)
    enc = O.fle_encode(vals, bw)
    c = synth.lt_constant(bw)
    row0, row1 = sharding.stripe_bounds(n_rows, world, rank)
    mine = sharding.stripe_word_slice(enc, bw, n_rows, world, rank)
    local = O.fle_pred(mine, row1 - row0, bw, O.OP_LT, c) if row1 > row0 else np.zeros(0, np.uint64)
    full = sharding.allgather_bitmap(torch.from_numpy(local.view(np.int64).copy()), n_rows, world)
    whole = O.fle_pred(enc, n_rows, bw, O.OP_LT, c)
    ok = np.array_equal(full.numpy().view(np.uint64), whole)
    # every rank sees the same gathered bitmap
    t = torch.tensor([int(ok)])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        ret.put(bool(t.item()))
    dist.destroy_process_group()


def _worker_q6(rank, world, port, n_rows, ret):
    """configs[4]: the three Q6 columns share the stripes; each rank ANDs its stripe's five leaf
    bitmaps (the oracle stands in for ips_eval_program) and the stripes' words are all-gathered."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as g
    from oracle import oracle as O
    pkg = g.load_package()
    sharding, q6 = pkg.sharding, pkg.q6
    ops = {"GE": O.OP_GE, "LT": O.OP_LT}
    row0, row1 = sharding.stripe_bounds(n_rows, world, rank)
    m = row1 - row0
    local = np.zeros(0, np.uint64)
    if m > 0:
        # rank r materialises only ITS rows of every column (generated from the row offset)
        encs = [O.fle_encode(q6.codes_numpy(c, m, start=row0), q6.COLUMNS[c][3]) for c in range(3)]
        for col, op, k in q6.LEAVES:
            leaf = O.fle_pred(encs[col], m, q6.COLUMNS[col][3], ops[op], k)
            local = leaf if local.size == 0 else (local & leaf)
    full = sharding.allgather_bitmap(torch.from_numpy(local.view(np.int64).copy()), n_rows, world)
    codes = [q6.codes_numpy(c, n_rows) for c in range(3)]
    exp = np.packbits(q6.truth(codes), bitorder="little")
    exp = np.concatenate([exp, np.zeros((-len(exp)) % 8, np.uint8)]).view(np.uint64)
    ok = np.array_equal(full.numpy().view(np.uint64), exp)
    t = torch.tensor([int(ok)])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        ret.put(bool(t.item()))
    dist.destroy_process_group()


def _worker_cyclic(rank, world, port, n_rows, n_chunks, ret):
    """Block-cyclic pieces + one all-gather per chunk (issued asynchronously, as bench.py overlaps
    them with the next chunk's scan): the gathered bitmap is in natural row order."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as g
    from oracle import oracle as O
    pkg = g.load_package()
    sharding, synth = pkg.sharding, pkg.synth
    bw = 9
    c = synth.lt_constant(bw)
    piece_rows, pieces = sharding.cyclic_pieces(n_rows, world, rank, n_chunks)
    pw = piece_rows // 64
    full = torch.zeros(n_chunks * world * pw, dtype=torch.int64)
    works = []
    for i, (row0, row1) in enumerate(pieces):
        m = row1 - row0
        local = np.zeros(0, np.uint64)
        if m > 0:
            enc = O.fle_encode(synth.column_u32(synth.SEED_HEADLINE, m, bw, start=row0), bw)
            local = O.fle_pred(enc, m, bw, O.OP_LT, c)
        works.append(sharding.allgather_chunk(torch.from_numpy(local.view(np.int64).copy()), full, i, pw, world))
    for w in works:
        w.wait()
    whole = O.fle_pred(O.fle_encode(synth.column_u32(synth.SEED_HEADLINE, n_rows, bw), bw), n_rows, bw, O.OP_LT, c)
    got = full.numpy().view(np.uint64)
    ok = np.array_equal(got[:len(whole)], whole) and not got[len(whole):].any()
    t = torch.tensor([int(ok)])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        ret.put(bool(t.item()))
    dist.destroy_process_group()


def _run_world(target, args, world=2):
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world) + args + (ret,)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return ret.get(timeout=10)


@pytest.mark.parametrize("n_rows,n_chunks", [(2048 * 40 + 13, 8), (5000, 8), (2048 * 16, 4)])
def test_block_cyclic_pieces_chunked_gather(n_rows, n_chunks):
    port = 33500 + (os.getpid() + n_rows) % 2000
    assert _run_world(_worker_cyclic, (port, n_rows, n_chunks)) is True


def test_cyclic_piece_geometry():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    sh = g.load_package().sharding
    for n in (1, 2049, 2048 * 64, 600_037_902):
        for world in (1, 2, 4, 8):
            for chunks in (1, 8):
                pr, _ = sh.cyclic_pieces(n, world, 0, chunks)
                assert pr % 2048 == 0 and pr * world * chunks >= n
                cover = []
                for i in range(chunks):
                    for r in range(world):
                        cover.append(sh.cyclic_pieces(n, world, r, chunks)[1][i])
                assert cover[0][0] == 0 and max(b for _, b in cover) == n
                for (a0, a1), (b0, b1) in zip(cover, cover[1:]):   # natural order, no gaps
                    assert a1 == b0 or (b0 == b1 == n)


@pytest.mark.parametrize("n_rows", [2048 * 5 + 77, 1500, 2048 * 4])
def test_q6_three_column_conjunction_over_stripes(n_rows):
    port = 31500 + (os.getpid() + n_rows) % 2000
    assert _run_world(_worker_q6, (port, n_rows)) is True


@pytest.mark.parametrize("n_rows,bw", [(10000, 12), (2048 * 7 + 5, 32), (100, 4), (2048 * 2, 9)])
def test_allgather_of_stripe_bitmaps(n_rows, bw):
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 29500 + (os.getpid() + n_rows) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_rows, bw, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(timeout=10) is True


def test_stripe_geometry():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    sh = g.load_package().sharding
    for n in (1, 2047, 2048, 2049, 600_037_902):
        for world in (1, 2, 4, 8):
            s = sh.stripe_rows(n, world)
            assert s % 2048 == 0 and s * world >= n
            bounds = [sh.stripe_bounds(n, world, r) for r in range(world)]
            assert bounds[0][0] == 0 and bounds[-1][1] == n
            for (a0, a1), (b0, b1) in zip(bounds, bounds[1:]):
                assert a1 == b0 and (a0 % 2048 == 0 or a0 == a1)

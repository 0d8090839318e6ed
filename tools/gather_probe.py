#!/usr/bin/env python3
"""The sharded step on a one-rank communicator, piece by piece: the un-chunked scan, the same scan
as ONE paged launch over 8 pieces (no signalling), and ips_fle_scan_allgather (signalling + waiters +
all-gathers).  For rocprofv3 --kernel-trace --stats: python3 tools/gather_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

ips = entry.load_package()
capi = ips.capi
dev = torch.device("cuda")
n, bw, n_chunks = 1 << 28, 32, 8
c = ips.synth.lt_constant(bw)
vals = capi.synth_u32(ips.synth.SEED_HEADLINE, n, bw)
enc = capi.fle_encode(vals, bw)
del vals
outs = capi.alloc_scan_outputs(n, dev)


def timeit(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return ts[len(ts) // 2], ts[0]


print("un-chunked scan", timeit(lambda: capi.fle_scan(enc, n, bw, capi.OP_LT, c, outputs=outs)), flush=True)
wpp = n // n_chunks // 64 * bw
chunk = capi.Chunk([(enc[i * wpp:(i + 1) * wpp], n // n_chunks, bw) for i in range(n_chunks)])
couts = chunk.alloc_outputs()
print("one paged launch over 8 pieces", timeit(lambda: chunk.fle_scan(capi.OP_LT, c, outputs=couts)), flush=True)
comm = capi.Comm(capi.comm_unique_id(), 1, 0)
full = torch.empty(n // 64, dtype=torch.int64, device=dev)
stream = torch.cuda.current_stream()


def step():
    comm.fle_scan_allgather(enc, n, bw, capi.OP_LT, c, n_chunks, outs[0], outs[1], outs[2], full, stream=stream)


def drained(reps=10):
    """the launch stream's share of a step (flag reset + the one scan launch) with the exchange of the
    step before it already finished: what the signalling costs the scan itself"""
    ts = []
    for _ in range(reps + 2):
        comm.join(stream)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        step()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts = sorted(ts[2:])
    return ts[len(ts) // 2], ts[0]


print("ips_fle_scan_allgather: scan launch of a step (exchange of the previous step drained)", drained(), flush=True)
print("ips_fle_scan_allgather (launch stream, back to back)", timeit(step), flush=True)
comm.join(stream)
torch.cuda.synchronize()
comm.check()
print("gathered == local:", torch.equal(full, outs[0][:n // 64]))
# whole steps including the exchange
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20):
    step()
comm.join(stream)
torch.cuda.synchronize()
print("step incl. exchange, us:", (time.perf_counter() - t0) / 20 * 1e6)
comm.close()

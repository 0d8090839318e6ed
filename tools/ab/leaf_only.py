#!/usr/bin/env python3
"""Dev tool for rocprofv3: 12 calls of the nullable leaf (w=12, 10 % NULL, 2^28 rows), nothing else."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as entry  # noqa: E402


def main():
    capi = entry.load_package().capi
    n, bw = 1 << 28, 12
    dev = torch.device("cuda")
    nn = capi.synth_u32(0x5EED0D1, n, 32)
    is_set = (nn.to(torch.int64) & 0xFFFFFFFF) >= int(0.1 * (1 << 32))
    defs = capi.fle_encode(is_set.to(torch.int32), 1)
    k = int(is_set.sum().item())
    del nn, is_set
    enc = capi.fle_encode(capi.synth_u32(0x5EED0D2, k, bw), bw)
    n_data = ((k + 63) // 64) * 64
    ws = capi.nullable_workspace(n, dev)
    bm = torch.empty((n + 63) // 64, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    for _ in range(12):
        capi.fle_pred_nullable(defs, 1, 1, n, enc, n_data, bw, capi.OP_LT, int(0.1 * (1 << bw)), bitmap=bm, workspace=ws)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()

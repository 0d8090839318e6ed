"""BASELINE configs[4]: TPC-H-Q6-shaped lineitem scan (SURVEY.md 8d "Config 5").

Three dictionary-coded FLE columns over N = 600,037,902 rows (SF100 lineitem), codes uniform per
column from the splitmix64 stream of its seed:

    l_shipdate  D = 2526 (w = 12)   Ge d0 AND Lt d1   one of seven years: codes [365, 730)
    l_discount  D = 11   (w = 4)    BETWEEN 0.05 AND 0.07: codes 5..7
    l_quantity  D = 50   (w = 6)    Lt 24: codes < 23

The conjunction is what HdfsParquetScanner::EvalSimplePredicates ANDs together
(hdfs-parquet-scanner.cc:1857-1862) after DictDecoder's literal -> code translation
(dict-encoding.h:461-541); here the leaves are already on codes.  Rows shard by stripes
(sharding.py); every column of the predicate uses the same stripes.
"""
import numpy as np

from . import synth

ROWS = 600_037_902
# name, seed, dictionary entries, code width
COLUMNS = (("l_shipdate", synth.SEED_Q6[0], 2526, 12),
           ("l_discount", synth.SEED_Q6[1], 11, 4),
           ("l_quantity", synth.SEED_Q6[2], 50, 6))
BITS_PER_ROW = sum(c[3] for c in COLUMNS)  # 22
# (column, op name, code) leaves of the conjunction, in evaluation order
LEAVES = ((0, "GE", 365), (0, "LT", 730), (1, "GE", 5), (1, "LT", 8), (2, "LT", 23))


def codes_numpy(col, n, start=0):
    """uint32 codes of rows [start, start + n) of column col."""
    _, seed, D, _ = COLUMNS[col]
    x = synth.splitmix64(seed, n, start) & np.uint64(0xFFFFFFFF)
    return (x % np.uint64(D)).astype(np.uint32)


def codes_gpu(capi, col, n, start=0, device=None):
    """The same codes generated on the GPU (ips_synth_splitmix_u32 + a modulo in torch: test /
    bench plumbing) as an int32 tensor."""
    import torch
    _, seed, D, _ = COLUMNS[col]
    x = capi.synth_u32(seed + start, n, 32, device=device)
    out = ((x.to(torch.int64) & 0xFFFFFFFF) % D).to(torch.int32)
    del x
    return out


def truth(codes):
    """Row model of the conjunction on three code arrays (numpy arrays or torch tensors)."""
    c0, c1, c2 = codes
    return (c0 >= 365) & (c0 < 730) & (c1 >= 5) & (c1 < 8) & (c2 < 23)


def program(capi, encs):
    """Postfix SimplePredicate program + column descriptors over the three encoded columns."""
    ops = {"GE": capi.OP_GE, "LT": capi.OP_LT}
    L, AND = capi.leaf, capi.and_node
    lv = [L(c, ops[o], k) for c, o, k in LEAVES]
    nodes = [lv[0], lv[1], AND(), lv[2], lv[3], AND(), AND(), lv[4], AND()]
    cols = [capi.fle_column(encs[i], COLUMNS[i][3]) for i in range(3)]
    return nodes, cols


def algorithmic_bytes(n_rows):
    """the columns' encoded bytes read once + ONE bitmap written."""
    words = (n_rows + 63) // 64
    return BITS_PER_ROW * 8 * words + 8 * words

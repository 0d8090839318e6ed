"""Deterministic synthetic columns: x_i = splitmix64(seed + i)."""
import numpy as np

SEED_HEADLINE = 0x5EED0001
SEED_DICT = 0x5EED0004
SEED_Q6 = (0x5EED0051, 0x5EED0052, 0x5EED0053)

_M64 = (1 << 64) - 1


def splitmix64(seed, n, start=0):
    """numpy twin of ips_synth_splitmix_u32's generator: returns uint64[n]."""
    with np.errstate(over="ignore"):
        z = (np.arange(start, start + n, dtype=np.uint64) + np.uint64(seed & _M64)
             + np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def column_u32(seed, n, bit_width, start=0):
    """v_i = x_i & (2^w - 1) as uint32 (taken from the low 32 bits of x_i)."""
    mask = np.uint64((1 << bit_width) - 1)
    return (splitmix64(seed, n, start) & np.uint64(0xFFFFFFFF) & mask).astype(np.uint32)


def lt_constant(bit_width, selectivity=0.10):
    """c = floor(sel * 2^w): LT c selects ~sel of uniform w-bit values."""
    return int(selectivity * (1 << bit_width))

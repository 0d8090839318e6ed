"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It wraps the plain-C restatement of the reference algorithms (oracle/fle_oracle.c) with numpy
arrays.  Nothing here is used by, or linked into, the HIP product path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "liboracle.so")

OP_EQ, OP_LT, OP_LE, OP_GT, OP_GE, OP_IN = range(6)
T_INT8, T_INT16, T_INT32, T_INT64, T_FLOAT, T_DOUBLE = range(6)
SEM_REFERENCE, SEM_SQL = 0, 1
XL_ALL_FALSE, XL_ALL_TRUE, XL_FLE = 0, 1, 2

NP_TYPES = {T_INT8: np.int8, T_INT16: np.int16, T_INT32: np.int32, T_INT64: np.int64,
            T_FLOAT: np.float32, T_DOUBLE: np.float64}


def build(force=False):
    """Compile the checker with gcc (sources are ours; the reference is not built)."""
    srcs = [os.path.join(_DIR, f) for f in ("fle_oracle.c", "cpu_baseline.c", "fle_oracle.h")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs)):
        return _SO
    subprocess.check_call(["make", "-C", _DIR, "-B", "liboracle.so"],
                          stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_fle_encoded_bytes.restype = C.c_int64
        _lib.orc_fle_encoded_bytes.argtypes = [C.c_int64, C.c_int]
        _lib.orc_dict_build.restype = C.c_int64
        _lib.orc_dict_write_data.restype = C.c_int64
        _lib.orc_skip_list.restype = C.c_int64
        _lib.orc_fle_select.restype = C.c_int64
        _lib.orc_bench_fused.restype = C.c_int64
        _lib.orc_bench_fused_best.restype = C.c_double
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def fle_encoded_bytes(n, bw):
    return int(lib().orc_fle_encoded_bytes(n, bw))


def bit_width_for_entries(d):
    return int(lib().orc_bit_width_for_entries(C.c_int64(d)))


def log2_ceil(x):
    return int(lib().orc_log2_ceil(C.c_uint64(x)))


def fle_encode(values, bw):
    v = np.ascontiguousarray(values, dtype=np.uint32)
    enc = np.zeros(fle_encoded_bytes(len(v), bw) // 8, dtype=np.uint64)
    lib().orc_fle_encode(_p(v), C.c_int64(len(v)), C.c_int(bw), _p(enc))
    return enc


def fle_decode(enc, n, bw):
    enc = np.ascontiguousarray(enc, dtype=np.uint64)
    out = np.zeros(n, dtype=np.uint32)
    lib().orc_fle_decode(_p(enc), C.c_int64(n), C.c_int(bw), _p(out))
    return out


def fle_pred(enc, n, bw, op, values):
    enc = np.ascontiguousarray(enc, dtype=np.uint64)
    vals = np.ascontiguousarray(np.atleast_1d(values), dtype=np.uint64)
    out = np.zeros((n + 63) // 64, dtype=np.uint64)
    lib().orc_fle_pred_words(_p(enc), C.c_int64(n), C.c_int(bw), C.c_int(op), _p(vals),
                             C.c_int(len(vals)), _p(out))
    return out


class Bitset:
    """dynamic_bitset behaviour used by the reference (append/push_back/resize/count)."""

    class _S(C.Structure):
        _fields_ = [("words", C.POINTER(C.c_uint64)), ("nbits", C.c_int64),
                    ("cap_words", C.c_int64)]

    def __init__(self):
        self.s = Bitset._S()
        lib().orc_bitset_init(C.byref(self.s))

    def __del__(self):
        try:
            lib().orc_bitset_free(C.byref(self.s))
        except Exception:
            pass

    def __len__(self):
        return int(self.s.nbits)

    def words(self):
        nw = (len(self) + 63) // 64
        return np.array([self.s.words[i] for i in range(nw)], dtype=np.uint64)

    def bits(self):
        w = self.words()
        return np.unpackbits(w.view(np.uint8), bitorder="little")[:len(self)].astype(bool)


class FleDecoder:
    """Stateful decoder with the reference cursor semantics (fle-encoding.h:344-567)."""

    class _S(C.Structure):
        _fields_ = [("buffer", C.c_void_p), ("buffer_end", C.c_void_p),
                    ("buffer_guard", C.c_void_p), ("bit_width", C.c_int), ("count", C.c_int),
                    ("current", C.c_uint32 * 64)]

    def __init__(self, enc, buffer_len, bw):
        self.enc = np.ascontiguousarray(enc, dtype=np.uint64)  # keep alive
        self.s = FleDecoder._S()
        lib().orc_fle_decoder_init(C.byref(self.s), _p(self.enc), C.c_int64(buffer_len),
                                   C.c_int(bw))

    def get(self):
        v = C.c_uint64()
        ok = lib().orc_fle_get(C.byref(self.s), C.byref(v))
        return bool(ok), int(v.value)

    def get_skip(self, skip):
        v = C.c_uint64()
        ok = lib().orc_fle_get_skip(C.byref(self.s), C.byref(v), C.c_int(skip))
        return bool(ok), int(v.value)

    def skip(self, skip):
        return bool(lib().orc_fle_skip(C.byref(self.s), C.c_int(skip)))

    def pred(self, op, num_rows, values, out=None):
        out = out or Bitset()
        vals = np.ascontiguousarray(np.atleast_1d(values), dtype=np.uint64)
        lib().orc_fle_pred(C.byref(self.s), C.c_int(op), C.c_int64(num_rows), C.byref(out.s),
                           _p(vals), C.c_int(len(vals)))
        return out


def dict_build(values, type_):
    """-> (sorted dictionary as np array of T, dictionary page bytes, codes u32)."""
    v = np.ascontiguousarray(values, dtype=NP_TYPES[type_])
    page_sz = 4 if type_ in (T_INT8, T_INT16) else v.itemsize
    page = np.zeros(max(len(v), 1) * page_sz, dtype=np.uint8)
    codes = np.zeros(len(v), dtype=np.uint32)
    d = lib().orc_dict_build(_p(v), C.c_int64(len(v)), C.c_int(type_), _p(page), _p(codes))
    if d < 0:
        raise ValueError("dictionary cap (40000) exceeded")
    page = page[:d * page_sz].copy()
    return dict_page_decode(page, type_), page, codes


def dict_page_decode(page, type_):
    """PLAIN dictionary page -> np array of T (DictDecoder ctor, dict-encoding.h:449-459)."""
    page = np.ascontiguousarray(page, dtype=np.uint8)
    if type_ in (T_INT8, T_INT16):
        return page.view(np.int32).astype(NP_TYPES[type_])
    return page.view(NP_TYPES[type_]).copy()


def dict_write_data(codes, num_entries):
    codes = np.ascontiguousarray(codes, dtype=np.uint32)
    bw = bit_width_for_entries(num_entries)
    page = np.zeros(1 + fle_encoded_bytes(len(codes), bw), dtype=np.uint8)
    n = lib().orc_dict_write_data(_p(codes), C.c_int64(len(codes)), C.c_int64(num_entries),
                                  _p(page))
    assert n == len(page)
    return page


def dict_translate(dict_arr, type_, op, literals):
    d = np.ascontiguousarray(dict_arr, dtype=NP_TYPES[type_])
    lit = np.ascontiguousarray(np.atleast_1d(literals), dtype=NP_TYPES[type_])
    fle_op = C.c_int(0)
    n_codes = C.c_int(0)
    codes = np.zeros(max(len(lit), 1), dtype=np.uint64)
    kind = lib().orc_dict_translate(_p(d), C.c_int64(len(d)), C.c_int(type_), C.c_int(op), _p(lit),
                                    C.c_int(len(lit)), C.byref(fle_op), _p(codes),
                                    C.byref(n_codes))
    return int(kind), int(fle_op.value), codes[:n_codes.value].copy()


def dict_pred(dict_arr, type_, data_page, n, op, literals):
    d = np.ascontiguousarray(dict_arr, dtype=NP_TYPES[type_])
    lit = np.ascontiguousarray(np.atleast_1d(literals), dtype=NP_TYPES[type_])
    page = np.ascontiguousarray(data_page, dtype=np.uint8)
    out = np.zeros((n + 63) // 64, dtype=np.uint64)
    lib().orc_dict_pred_words(_p(d), C.c_int64(len(d)), C.c_int(type_), _p(page),
                              C.c_int64(len(page)), C.c_int64(n), C.c_int(op), _p(lit),
                              C.c_int(len(lit)), _p(out))
    return out


def dict_decode(dict_arr, type_, data_page, n):
    d = np.ascontiguousarray(dict_arr, dtype=NP_TYPES[type_])
    page = np.ascontiguousarray(data_page, dtype=np.uint8)
    out = np.zeros(n, dtype=NP_TYPES[type_])
    ok = lib().orc_dict_decode(_p(d), C.c_int64(len(d)), C.c_int(type_), _p(page),
                               C.c_int64(len(page)), C.c_int64(n), _p(out))
    return bool(ok), out


def plain_encode(values, type_):
    """PLAIN page bytes: 4-byte slots for int8/16/32/float, 8 for int64/double."""
    v = np.ascontiguousarray(values, dtype=NP_TYPES[type_])
    if type_ in (T_INT8, T_INT16):
        return v.astype(np.int32).view(np.uint8).copy()
    return v.view(np.uint8).copy()


def plain_pred(page, n, type_, op, literals, semantics):
    page = np.ascontiguousarray(page, dtype=np.uint8)
    lit = np.ascontiguousarray(np.atleast_1d(literals), dtype=NP_TYPES[type_])
    out = np.zeros((n + 63) // 64, dtype=np.uint64)
    lib().orc_plain_pred_words(_p(page), C.c_int64(n), C.c_int(type_), C.c_int(op), _p(lit),
                               C.c_int(len(lit)), C.c_int(semantics), _p(out))
    return out


def bitmap_expand(root, sub, n_rows):
    root = np.ascontiguousarray(root, dtype=np.uint64)
    sub = np.ascontiguousarray(sub, dtype=np.uint64)
    out = np.zeros((n_rows + 63) // 64, dtype=np.uint64)
    lib().orc_bitmap_expand(_p(root), _p(sub), C.c_int64(n_rows), _p(out))
    return out


def skip_list(bitmap, n_rows):
    bm = np.ascontiguousarray(bitmap, dtype=np.uint64)
    skips = np.zeros(max(n_rows, 1), dtype=np.int32)
    last = C.c_int64(0)
    cnt = lib().orc_skip_list(_p(bm), C.c_int64(n_rows), _p(skips), C.byref(last))
    return skips[:cnt].copy(), int(last.value)


def fle_select(enc, n, bw, bitmap):
    enc = np.ascontiguousarray(enc, dtype=np.uint64)
    bm = np.ascontiguousarray(bitmap, dtype=np.uint64)
    out = np.zeros(max(n, 1), dtype=np.uint32)
    cnt = lib().orc_fle_select(_p(enc), C.c_int64(enc.nbytes), C.c_int64(n), C.c_int(bw), _p(bm),
                               _p(out))
    if cnt < 0:
        raise RuntimeError("oracle: ran out of data")
    return out[:cnt].copy()


def fast_unpack_block(blk, bw):
    blk = np.ascontiguousarray(blk, dtype=np.uint64)
    out = np.zeros(64, dtype=np.uint32)
    lib().orc_fast_unpack_block(_p(blk), C.c_int(bw), _p(out))
    return out


def bench_fused(enc, n, bw, op, value, threads, mode=0):
    """Timed CPU baseline body; returns (n_selected, bitmap, sel_out)."""
    enc = np.ascontiguousarray(enc, dtype=np.uint64)
    bitmap = np.zeros((n + 63) // 64, dtype=np.uint64)
    sel = np.empty(n, dtype=np.uint32)
    cnt = lib().orc_bench_fused(_p(enc), C.c_int64(n), C.c_int(bw), C.c_int(op),
                                C.c_uint64(value), C.c_int(threads), C.c_int(mode), _p(bitmap),
                                _p(sel))
    return int(cnt), bitmap, sel


def bench_buffers(n):
    """Output buffers of the timed baseline, allocated and touched ONCE by the caller: a fresh
    np.empty(n) per run would add a page fault per 4 KiB of output to every timed call."""
    bitmap = np.zeros((n + 63) // 64, dtype=np.uint64)
    sel = np.zeros(n, dtype=np.uint32)
    return bitmap, sel


def bench_fused_best(enc, n, bw, op, value, threads, mode, buffers, reps=5):
    """Best-of-reps seconds of the baseline body, timed inside C after one warm-up pass.
    -> (seconds, n_selected)"""
    enc = np.ascontiguousarray(enc, dtype=np.uint64)
    bitmap, sel = buffers
    assert len(bitmap) >= (n + 63) // 64 and len(sel) >= n
    cnt = C.c_int64(0)
    sec = lib().orc_bench_fused_best(_p(enc), C.c_int64(n), C.c_int(bw), C.c_int(op),
                                     C.c_uint64(value), C.c_int(threads), C.c_int(mode),
                                     _p(bitmap), _p(sel), C.c_int(reps), C.byref(cnt))
    return float(sec), int(cnt.value)


def hw_threads():
    return int(lib().orc_hw_threads())


def has_avx2():
    return bool(lib().orc_has_avx2())

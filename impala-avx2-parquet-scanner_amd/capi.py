"""ctypes binding of libips_hip.so (include/ips.h) for tests and bench.py.

torch is used only as plumbing: device buffers, streams, torch.distributed.  Every function here
is a thin call through the C-ABI; there is no Python or CPU fallback -- a missing or failing
library raises.

torch must be imported before the library is loaded so that both share one HIP runtime.
"""
import ctypes as C
import os

import numpy as np
import torch  # noqa: F401  (load torch's HIP runtime first)

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IPS_LIB") or os.path.join(_DIR, "libips_hip.so")  # IPS_LIB: dev builds

OP_EQ, OP_LT, OP_LE, OP_GT, OP_GE, OP_IN = range(6)
T_INT8, T_INT16, T_INT32, T_INT64, T_FLOAT, T_DOUBLE = range(6)
SEM_REFERENCE, SEM_SQL = 0, 1
XL_ALL_FALSE, XL_ALL_TRUE, XL_FLE = 0, 1, 2
NODE_LEAF, NODE_AND, NODE_OR = 0, 1, 2
COL_FLE, COL_PLAIN = 0, 1
BATCH_ROWS = 2048
MAX_IN_LIST = 256

NP_TYPES = {T_INT8: np.int8, T_INT16: np.int16, T_INT32: np.int32, T_INT64: np.int64,
            T_FLOAT: np.float32, T_DOUBLE: np.float64}
TORCH_SLOT = {T_INT8: torch.int32, T_INT16: torch.int32, T_INT32: torch.int32,
              T_INT64: torch.int64, T_FLOAT: torch.float32, T_DOUBLE: torch.float64}

# every symbol include/ips.h declares (tests/test_abi.py checks the header against this list and
# the built library against both)
SYMBOLS = [
    "ips_version", "ips_last_error", "ips_device_count", "ips_set_device", "ips_device_info",
    "ips_malloc", "ips_free", "ips_memcpy_h2d", "ips_memcpy_d2h", "ips_memset",
    "ips_stream_create", "ips_stream_destroy", "ips_stream_synchronize",
    "ips_fle_encoded_bytes", "ips_fle_encode", "ips_fle_decode", "ips_fle_pred", "ips_fle_scan",
    "ips_fle_select", "ips_fle_scan_pages", "ips_batches_workspace_bytes", "ips_batches_compact", "ips_assemble_tuples",
    "ips_assemble_workspace_bytes", "ips_bitmap_compress",
    "ips_dict_open", "ips_dict_close", "ips_dict_num_entries", "ips_dict_bit_width", "ips_dict_encode",
    "ips_dict_encode_workspace_bytes", "ips_program_workspace_bytes",
    "ips_nullable_workspace_bytes", "ips_fle_pred_nullable", "ips_dict_pred_nullable",
    "ips_select_nullable_workspace_bytes", "ips_dict_select_nullable",
    "ips_dict_translate", "ips_dict_pred", "ips_dict_decode", "ips_dict_scan", "ips_dict_select",
    "ips_plain_stride", "ips_plain_pred", "ips_plain_pred_nullable", "ips_plain_scan", "ips_plain_select",
    "ips_bitmap_and", "ips_bitmap_or", "ips_bitmap_fill", "ips_bitmap_count", "ips_bitmap_batch_counts",
    "ips_expand_workspace_bytes", "ips_bitmap_expand",
    "ips_eval_program", "ips_set_program_strategy", "ips_synth_splitmix_u32",
    "ips_chunk_select_nullable", "ips_chunk_select_nullable_workspace_bytes",
    "ips_inset_open", "ips_dict_inset_open", "ips_inset_close", "ips_inset_size", "ips_fle_pred_inset",
    "ips_fle_scan_inset", "ips_dict_scan_inset",
    "ips_chunk_open", "ips_chunk_close", "ips_chunk_num_rows", "ips_chunk_num_batches", "ips_chunk_num_pages",
    "ips_chunk_program_workspace_bytes", "ips_eval_program_chunks", "ips_chunk_fle_scan", "ips_chunk_dict_scan",
    "ips_chunk_plain_scan", "ips_chunk_select",
    "ips_comm_unique_id", "ips_comm_init", "ips_comm_destroy", "ips_allgather_bitmap",
    "ips_fle_scan_allgather", "ips_comm_join", "ips_eval_program_chunks_allgather", "ips_comm_check",
]


class IpsError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"ips status {status}: {msg}")
        self.status = status


class Column(C.Structure):
    _fields_ = [("encoding", C.c_int32), ("bit_width", C.c_int32), ("type", C.c_int32),
                ("max_def_level", C.c_int32), ("d_data", C.c_void_p), ("d_def_levels", C.c_void_p),
                ("def_bit_width", C.c_int32), ("reserved", C.c_int32), ("n_data_rows", C.c_int64)]


class TupleColumn(C.Structure):
    _fields_ = [("d_batch_values", C.c_void_p), ("value_width", C.c_int32),
                ("tuple_offset", C.c_int32), ("d_dense_values", C.c_void_p),
                ("d_nonnull_flags", C.c_void_p), ("null_byte_offset", C.c_int32),
                ("null_bit_mask", C.c_int32)]


class Node(C.Structure):
    _fields_ = [("kind", C.c_int32), ("column", C.c_int32), ("op", C.c_int32),
                ("n_consts", C.c_int32), ("consts", C.c_uint64 * 16), ("inset", C.c_void_p)]


_lib = None


def lib():
    """Load libips_hip.so; raises if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.ips_last_error.restype = C.c_char_p
        L.ips_fle_encoded_bytes.restype = C.c_int64
        L.ips_fle_encoded_bytes.argtypes = [C.c_int64, C.c_int]
        L.ips_dict_num_entries.restype = C.c_int64
        L.ips_dict_num_entries.argtypes = [C.c_void_p]
        L.ips_dict_bit_width.argtypes = [C.c_int64]
        L.ips_program_workspace_bytes.restype = C.c_size_t
        L.ips_program_workspace_bytes.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64]
        for name in ("ips_batches_workspace_bytes", "ips_expand_workspace_bytes",
                     "ips_dict_encode_workspace_bytes", "ips_nullable_workspace_bytes"):
            getattr(L, name).restype = C.c_size_t
            getattr(L, name).argtypes = [C.c_int64]
        _lib = L
    return _lib


def _ck(status):
    if status != 0:
        raise IpsError(status, lib().ips_last_error().decode())


def _ptr(t):
    """device pointer of a torch tensor (or pass through an int / None)."""
    if t is None:
        return C.c_void_p(0)
    if isinstance(t, torch.Tensor):
        assert t.is_contiguous()
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(int(t))


def _stream(stream=None):
    if stream is None:
        stream = torch.cuda.current_stream()
    return C.c_void_p(stream.cuda_stream)


def _words(n_rows):
    return (n_rows + 63) // 64


def n_batches(n_rows):
    return (n_rows + BATCH_ROWS - 1) // BATCH_ROWS


def version():
    return int(lib().ips_version())


def device_info():
    name = C.create_string_buffer(128)
    cus = C.c_int(0)
    hbm = C.c_int64(0)
    _ck(lib().ips_device_info(name, 128, C.byref(cus), C.byref(hbm)))
    return name.value.decode(), cus.value, hbm.value


def fle_encoded_bytes(n_rows, bw):
    return int(lib().ips_fle_encoded_bytes(n_rows, bw))


def _consts(values):
    v = np.ascontiguousarray(np.atleast_1d(values), dtype=np.uint64)
    return v, v.ctypes.data_as(C.POINTER(C.c_uint64)), len(v)


# ---- FLE ------------------------------------------------------------------------------------
def fle_encode(values, bw, out=None, stream=None):
    """values: cuda tensor of uint8 / int16 / int32 (unsigned meaning) -> int64 words."""
    in_width = values.element_size()
    n = values.numel()
    if out is None:
        out = torch.empty(max(fle_encoded_bytes(n, bw) // 8, 2), dtype=torch.int64,
                          device=values.device)
    _ck(lib().ips_fle_encode(_ptr(values), in_width, C.c_int64(n), bw, _ptr(out), _stream(stream)))
    return out[:fle_encoded_bytes(n, bw) // 8]


def fle_decode(enc, n_rows, bw, out_width=4, out=None, stream=None):
    dt = {1: torch.uint8, 2: torch.int16, 4: torch.int32}[out_width]
    if out is None:
        out = torch.empty(max(n_rows, 16), dtype=dt, device=enc.device)
    _ck(lib().ips_fle_decode(_ptr(enc), C.c_int64(n_rows), bw, _ptr(out), out_width,
                             _stream(stream)))
    return out[:n_rows]


def fle_pred(enc, n_rows, bw, op, values, bitmap=None, stream=None):
    keep, p, k = _consts(values)
    if bitmap is None:
        bitmap = torch.empty(max(_words(n_rows), 2), dtype=torch.int64, device=enc.device)
    _ck(lib().ips_fle_pred(_ptr(enc), C.c_int64(n_rows), bw, op, p, k, _ptr(bitmap),
                           _stream(stream)))
    return bitmap[:_words(n_rows)]


def nullable_workspace(n_rows, device):
    return torch.empty(max(int(lib().ips_nullable_workspace_bytes(n_rows)), 16), dtype=torch.uint8,
                       device=device)


def fle_pred_nullable(def_levels, def_bw, max_def, n_rows, data_enc, n_data_rows, bw, op, values,
                      bitmap=None, workspace=None, stream=None):
    """Predicate on an OPTIONAL FLE column: def levels + data blocks -> bitmap over all rows."""
    keep, p, k = _consts(values)
    if bitmap is None:
        bitmap = torch.empty(max(_words(n_rows), 2), dtype=torch.int64, device=def_levels.device)
    if workspace is None:
        workspace = nullable_workspace(n_rows, def_levels.device)
    _ck(lib().ips_fle_pred_nullable(_ptr(def_levels), def_bw, max_def, C.c_int64(n_rows),
                                    _ptr(data_enc), C.c_int64(n_data_rows), bw, op, p, k,
                                    _ptr(bitmap), _ptr(workspace), _stream(stream)))
    return bitmap[:_words(n_rows)]


def alloc_scan_outputs(n_rows, device, value_dtype=torch.int32):
    nb = max(n_batches(n_rows), 1)
    bitmap = torch.empty(max(_words(n_rows), 2), dtype=torch.int64, device=device)
    values = torch.empty(nb * BATCH_ROWS, dtype=value_dtype, device=device)
    counts = torch.empty(nb, dtype=torch.int32, device=device)
    return bitmap, values, counts


def fle_scan(enc, n_rows, bw, op, values, outputs=None, stream=None):
    """-> (bitmap words, batch values, batch counts)"""
    keep, p, k = _consts(values)
    bitmap, bvals, counts = outputs or alloc_scan_outputs(n_rows, enc.device)
    _ck(lib().ips_fle_scan(_ptr(enc), C.c_int64(n_rows), bw, op, p, k, _ptr(bitmap), _ptr(bvals),
                           _ptr(counts), _stream(stream)))
    return bitmap[:_words(n_rows)], bvals, counts[:n_batches(n_rows)]


class PageScan(C.Structure):
    _fields_ = [("d_enc", C.c_void_p), ("n_rows", C.c_int64), ("d_bitmap", C.c_void_p),
                ("d_batch_values", C.c_void_p), ("d_batch_counts", C.c_void_p)]


def make_page_list(pages):
    """pages: list of (enc tensor, n_rows, (bitmap, batch_values, batch_counts)).  -> ctypes array
    (build it once per column chunk; ips_fle_scan_pages reads it on the host at every call)."""
    arr = (PageScan * len(pages))()
    for i, (enc, n_rows, outs) in enumerate(pages):
        arr[i].d_enc = enc.data_ptr()
        arr[i].n_rows = n_rows
        arr[i].d_bitmap, arr[i].d_batch_values, arr[i].d_batch_counts = (t.data_ptr() for t in outs)
    return arr


def fle_scan_pages(page_list, bw, op, values, stream=None):
    """ips_fle_scan over a list of separate pages (make_page_list) in ceil(n/64) launches."""
    keep, p, k = _consts(values)
    _ck(lib().ips_fle_scan_pages(page_list, len(page_list), bw, op, p, k, _stream(stream)))


def fle_select(enc, n_rows, bw, bitmap, outputs=None, stream=None):
    _, bvals, counts = outputs or alloc_scan_outputs(n_rows, enc.device)
    _ck(lib().ips_fle_select(_ptr(enc), C.c_int64(n_rows), bw, _ptr(bitmap), _ptr(bvals),
                             _ptr(counts), _stream(stream)))
    return bvals, counts[:n_batches(n_rows)]


def batches_compact(bvals, counts, n_rows, stream=None):
    """-> dense tensor of the selected values in row order."""
    width = bvals.element_size()
    ws = torch.empty(max(int(lib().ips_batches_workspace_bytes(n_rows)), 16), dtype=torch.uint8,
                     device=bvals.device)
    dense = torch.empty(max(n_rows, 16), dtype=bvals.dtype, device=bvals.device)
    total = torch.zeros(1, dtype=torch.int64, device=bvals.device)
    _ck(lib().ips_batches_compact(_ptr(bvals), _ptr(counts), C.c_int64(n_rows), width,
                                  _ptr(dense), _ptr(total), _ptr(ws), _stream(stream)))
    return dense[:int(total.item())]


def assemble_tuples(columns, counts, n_rows, tuple_size, template=None, stream=None):
    """columns: list of (batch_values tensor, tuple_offset) for REQUIRED columns or
    (dense_values, tuple_offset, nonnull_flags, null_byte_offset, null_bit_mask) for OPTIONAL
    ones, all selected by the bitmap behind 'counts'.  template: tuple_size bytes every tuple
    starts from (None = zeros).  Returns a uint8 tensor [n_tuples, tuple_size] of row-major tuples
    in row order."""
    dev = counts.device
    arr = (TupleColumn * len(columns))()
    n_opt = 0
    for i, col in enumerate(columns):
        bv, off = col[0], col[1]
        arr[i].value_width = bv.element_size()
        arr[i].tuple_offset = off
        if len(col) > 2:
            n_opt += 1
            arr[i].d_dense_values = bv.data_ptr()
            arr[i].d_nonnull_flags = col[2].data_ptr()
            arr[i].null_byte_offset, arr[i].null_bit_mask = col[3], col[4]
        else:
            arr[i].d_batch_values = bv.data_ptr()
    lib().ips_assemble_workspace_bytes.restype = C.c_size_t
    ws = torch.empty(int(lib().ips_assemble_workspace_bytes(C.c_int64(n_rows), n_opt)) + 16,
                     dtype=torch.uint8, device=dev)
    total = torch.zeros(1, dtype=torch.int64, device=dev)
    cap = int(counts.to(torch.int64).sum().item())
    tuples = torch.full((max(cap, 1) * tuple_size + 16,), 0xA5, dtype=torch.uint8, device=dev)
    tmpl = None
    if template is not None:
        tmpl = np.ascontiguousarray(template, dtype=np.uint8)
        assert len(tmpl) == tuple_size
    _ck(lib().ips_assemble_tuples(arr, len(columns), _ptr(counts), C.c_int64(n_rows), tuple_size,
                                  tmpl.ctypes.data_as(C.c_void_p) if tmpl is not None else None,
                                  _ptr(tuples), _ptr(total), _ptr(ws), _stream(stream)))
    n = int(total.item())
    return tuples[:n * tuple_size].view(n, tuple_size)


# ---- dictionary -----------------------------------------------------------------------------
class Dict:
    """ips_dict handle (DictDecoder<T>, dict-encoding.h:202-232)."""

    def __init__(self, dict_page, type_):
        page = np.ascontiguousarray(dict_page, dtype=np.uint8)
        self.type = type_
        self.h = C.c_void_p(0)
        _ck(lib().ips_dict_open(page.ctypes.data_as(C.c_void_p), C.c_int64(len(page)), type_,
                                C.byref(self.h)))

    def close(self):
        if self.h:
            lib().ips_dict_close(self.h)
            self.h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def num_entries(self):
        return int(lib().ips_dict_num_entries(self.h))

    def _lits(self, literals):
        v = np.ascontiguousarray(np.atleast_1d(literals), dtype=NP_TYPES[self.type])
        return v, v.ctypes.data_as(C.c_void_p), len(v)

    def translate(self, op, literals):
        keep, p, k = self._lits(literals)
        kind, fle_op, n_codes = C.c_int(0), C.c_int(0), C.c_int(0)
        codes = (C.c_uint64 * max(k, 1))()
        _ck(lib().ips_dict_translate(self.h, op, p, k, C.byref(kind), C.byref(fle_op), codes,
                                     C.byref(n_codes)))
        return kind.value, fle_op.value, [int(codes[i]) for i in range(n_codes.value)]

    def pred(self, codes_enc, n_rows, bw, op, literals, bitmap=None, stream=None):
        keep, p, k = self._lits(literals)
        if bitmap is None:
            bitmap = torch.empty(max(_words(n_rows), 2), dtype=torch.int64, device=codes_enc.device)
        _ck(lib().ips_dict_pred(self.h, _ptr(codes_enc), C.c_int64(n_rows), bw, op, p, k,
                                _ptr(bitmap), _stream(stream)))
        return bitmap[:_words(n_rows)]

    def decode(self, codes_enc, n_rows, bw, stream=None):
        out = torch.empty(max(n_rows, 16), dtype=TORCH_SLOT[self.type], device=codes_enc.device)
        bad = torch.zeros(1, dtype=torch.int32, device=codes_enc.device)
        _ck(lib().ips_dict_decode(self.h, _ptr(codes_enc), C.c_int64(n_rows), bw, _ptr(out),
                                  _ptr(bad), _stream(stream)))
        return out[:n_rows], bad

    def scan(self, codes_enc, n_rows, bw, op, literals, stream=None):
        keep, p, k = self._lits(literals)
        bitmap, bvals, counts = alloc_scan_outputs(n_rows, codes_enc.device, TORCH_SLOT[self.type])
        _ck(lib().ips_dict_scan(self.h, _ptr(codes_enc), C.c_int64(n_rows), bw, op, p, k,
                                _ptr(bitmap), _ptr(bvals), _ptr(counts), _stream(stream)))
        return bitmap[:_words(n_rows)], bvals, counts[:n_batches(n_rows)]


def _dict_pred_nullable(self, def_levels, def_bw, max_def, n_rows, codes_enc, n_data_rows, bw, op,
                        literals, bitmap=None, workspace=None, stream=None):
    keep, p, k = self._lits(literals)
    if bitmap is None:
        bitmap = torch.empty(max(_words(n_rows), 2), dtype=torch.int64, device=def_levels.device)
    if workspace is None:
        workspace = nullable_workspace(n_rows, def_levels.device)
    _ck(lib().ips_dict_pred_nullable(self.h, _ptr(def_levels), def_bw, max_def, C.c_int64(n_rows),
                                     _ptr(codes_enc), C.c_int64(n_data_rows), bw, op, p, k,
                                     _ptr(bitmap), _ptr(workspace), _stream(stream)))
    return bitmap[:_words(n_rows)]


Dict.pred_nullable = _dict_pred_nullable


def select_nullable(dict_, def_levels, def_bw, max_def, n_rows, codes_enc, n_data_rows, bw, selection, stream=None):
    """One-call late materialisation of an OPTIONAL column (dict_ may be None: raw FLE values).
    -> (dense values of the selected non-NULL rows, NOT-NULL flag words per selected row,
        n_selected, n_selected_non_null)"""
    dev = def_levels.device
    vw = 4 if dict_ is None else torch.empty(0, dtype=TORCH_SLOT[dict_.type]).element_size()
    lib().ips_select_nullable_workspace_bytes.restype = C.c_size_t
    ws = torch.empty(int(lib().ips_select_nullable_workspace_bytes(C.c_int64(n_rows), C.c_int64(n_data_rows), vw)) + 16,
                     dtype=torch.uint8, device=dev)
    dense = torch.empty(max(n_data_rows, 16), dtype=torch.int32 if dict_ is None else TORCH_SLOT[dict_.type], device=dev)
    flags = torch.empty(max(_words(n_rows), 2), dtype=torch.int64, device=dev)
    counts = torch.zeros(3, dtype=torch.int64, device=dev)
    _ck(lib().ips_dict_select_nullable(dict_.h if dict_ is not None else None, _ptr(def_levels), def_bw, max_def,
                                       C.c_int64(n_rows), _ptr(codes_enc), C.c_int64(n_data_rows), bw,
                                       _ptr(selection), _ptr(dense), _ptr(flags), _ptr(counts), _ptr(ws),
                                       _stream(stream)))
    n_sel, n_val, bad = (int(x) for x in counts.cpu().tolist())
    if bad:
        raise IpsError(6, "a selected code lies outside the dictionary (IPS_ERR_BAD_INDEX)")
    return dense[:n_val], flags[:_words(max(n_sel, 1))], n_sel, n_val


def _dict_select(self, codes_enc, n_rows, bw, bitmap, stream=None):
    _, bvals, counts = alloc_scan_outputs(n_rows, codes_enc.device, TORCH_SLOT[self.type])
    _ck(lib().ips_dict_select(self.h, _ptr(codes_enc), C.c_int64(n_rows), bw, _ptr(bitmap),
                              _ptr(bvals), _ptr(counts), _stream(stream)))
    return bvals, counts[:n_batches(n_rows)]


Dict.select = _dict_select


def dict_encode(values, type_, stream=None):
    """values: cuda tensor of PLAIN slots (int32/float32/int64/float64).  -> (dictionary page
    bytes (np.uint8), code bit width, FLE blocks of the codes (int64 words))."""
    n = values.numel()
    page = np.zeros(40000 * 8, dtype=np.uint8)
    dict_len = C.c_int64(0)
    bw = C.c_int(0)
    enc = torch.empty(max(fle_encoded_bytes(n, 16) // 8, 2), dtype=torch.int64, device=values.device)
    ws = torch.empty(int(lib().ips_dict_encode_workspace_bytes(n)), dtype=torch.uint8,
                     device=values.device)
    _ck(lib().ips_dict_encode(_ptr(values), C.c_int64(n), type_, page.ctypes.data_as(C.c_void_p),
                              C.c_int64(len(page)), C.byref(dict_len), C.byref(bw), _ptr(enc),
                              _ptr(ws), _stream(stream)))
    return page[:dict_len.value].copy(), bw.value, enc[:fle_encoded_bytes(n, bw.value) // 8]


def dict_bit_width(num_entries):
    return int(lib().ips_dict_bit_width(num_entries))


# ---- PLAIN ----------------------------------------------------------------------------------
def plain_pred(page, n_rows, type_, op, literals, semantics=SEM_SQL, bitmap=None, stream=None):
    v = np.ascontiguousarray(np.atleast_1d(literals), dtype=NP_TYPES[type_])
    if bitmap is None:
        bitmap = torch.empty(max(_words(n_rows), 2), dtype=torch.int64, device=page.device)
    _ck(lib().ips_plain_pred(_ptr(page), C.c_int64(n_rows), type_, op,
                             v.ctypes.data_as(C.c_void_p), len(v), semantics, _ptr(bitmap),
                             _stream(stream)))
    return bitmap[:_words(n_rows)]


def plain_pred_nullable(def_levels, def_bw, max_def, n_rows, page, n_data_rows, type_, op, literals, bitmap=None,
                        workspace=None, stream=None):
    """ips_plain_pred_nullable: SQL semantics on an OPTIONAL PLAIN page (stored values + levels)."""
    v = np.ascontiguousarray(np.atleast_1d(literals), dtype=NP_TYPES[type_])
    if bitmap is None:
        bitmap = torch.empty(max(_words(n_rows), 2), dtype=torch.int64, device=def_levels.device)
    if workspace is None:
        workspace = nullable_workspace(n_rows, def_levels.device)
    _ck(lib().ips_plain_pred_nullable(_ptr(def_levels), def_bw, max_def, C.c_int64(n_rows), _ptr(page),
                                      C.c_int64(n_data_rows), type_, op, v.ctypes.data_as(C.c_void_p), len(v),
                                      _ptr(bitmap), _ptr(workspace), _stream(stream)))
    return bitmap[:_words(n_rows)]


def nullable_plain_column(def_levels, def_bw, max_def, page, type_, n_data_rows):
    """An OPTIONAL PLAIN column of a program: levels of all rows + the stored (non-NULL) values."""
    c = Column()
    c.encoding, c.bit_width, c.type, c.d_data = COL_PLAIN, 0, type_, page.data_ptr()
    c.max_def_level, c.d_def_levels, c.def_bit_width, c.n_data_rows = max_def, def_levels.data_ptr(), def_bw, n_data_rows
    return c


# ---- bitmap algebra -------------------------------------------------------------------------
def bitmap_and(a, b, n_rows, stream=None):
    _ck(lib().ips_bitmap_and(_ptr(a), _ptr(b), C.c_int64(n_rows), _stream(stream)))
    return a


def bitmap_or(a, b, n_rows, stream=None):
    _ck(lib().ips_bitmap_or(_ptr(a), _ptr(b), C.c_int64(n_rows), _stream(stream)))
    return a


def bitmap_fill(a, n_rows, value, stream=None):
    _ck(lib().ips_bitmap_fill(_ptr(a), C.c_int64(n_rows), int(value), _stream(stream)))
    return a


def bitmap_count(a, n_rows, stream=None):
    cnt = torch.zeros(1, dtype=torch.int64, device=a.device)
    _ck(lib().ips_bitmap_count(_ptr(a), C.c_int64(n_rows), _ptr(cnt), _stream(stream)))
    return int(cnt.item())


def bitmap_batch_counts(a, n_rows, stream=None):
    counts = torch.empty(max(n_batches(n_rows), 1), dtype=torch.int32, device=a.device)
    _ck(lib().ips_bitmap_batch_counts(_ptr(a), C.c_int64(n_rows), _ptr(counts), _stream(stream)))
    return counts[:n_batches(n_rows)]


def bitmap_compress(mask, src, n_rows, stream=None):
    """-> (out words, popcount(mask)): out bit j = src at the j-th set bit of mask."""
    ws = torch.empty(max(int(lib().ips_expand_workspace_bytes(n_rows)), 16), dtype=torch.uint8,
                     device=mask.device)
    out = torch.empty(max(_words(n_rows), 2), dtype=torch.int64, device=mask.device)
    n_out = torch.zeros(1, dtype=torch.int64, device=mask.device)
    _ck(lib().ips_bitmap_compress(_ptr(mask), _ptr(src), C.c_int64(n_rows), _ptr(out),
                                  _ptr(n_out), _ptr(ws), _stream(stream)))
    k = int(n_out.item())
    return out[:_words(n_rows)], k


def bitmap_expand(root, sub, n_rows, stream=None):
    ws = torch.empty(max(int(lib().ips_expand_workspace_bytes(n_rows)), 16), dtype=torch.uint8,
                     device=root.device)
    out = torch.empty(max(_words(n_rows), 2), dtype=torch.int64, device=root.device)
    _ck(lib().ips_bitmap_expand(_ptr(root), _ptr(sub), C.c_int64(n_rows), _ptr(out), _ptr(ws),
                                _stream(stream)))
    return out[:_words(n_rows)]


def plain_scan(page, n_rows, type_, op, literals, semantics=None, op2=None, literal2=None, stream=None):
    """-> (bitmap words, batch slots, batch counts): predicate and selected slots in one pass."""
    stride = int(lib().ips_plain_stride(type_))
    dt = torch.int32 if stride == 4 else torch.int64
    bitmap, bvals, counts = alloc_scan_outputs(n_rows, page.device, dt)
    v = np.ascontiguousarray(np.atleast_1d(literals), dtype=NP_TYPES[type_])
    v2 = None if literal2 is None else np.ascontiguousarray(np.atleast_1d(literal2), dtype=NP_TYPES[type_])
    _ck(lib().ips_plain_scan(_ptr(page), C.c_int64(n_rows), type_, op, v.ctypes.data_as(C.c_void_p),
                             len(v), 0 if op2 is None else op2,
                             None if v2 is None else v2.ctypes.data_as(C.c_void_p),
                             SEM_SQL if semantics is None else semantics, _ptr(bitmap), _ptr(bvals),
                             _ptr(counts), _stream(stream)))
    return bitmap[:_words(n_rows)], bvals, counts[:n_batches(n_rows)]


def plain_select(page, n_rows, type_, bitmap, stream=None):
    """-> (batch values as 4- or 8-byte integer slots, batch counts) of the rows set in bitmap."""
    stride = int(lib().ips_plain_stride(type_))
    dt = torch.int32 if stride == 4 else torch.int64
    _, bvals, counts = alloc_scan_outputs(n_rows, page.device, dt)
    _ck(lib().ips_plain_select(_ptr(page), C.c_int64(n_rows), type_, _ptr(bitmap), _ptr(bvals),
                               _ptr(counts), _stream(stream)))
    return bvals, counts[:n_batches(n_rows)]


# ---- fused predicate program ----------------------------------------------------------------
def leaf(column, op, consts):
    n = Node()
    n.kind, n.column, n.op = NODE_LEAF, column, op
    cs = np.atleast_1d(consts)
    n.n_consts = len(cs)
    for i, c in enumerate(cs):
        n.consts[i] = int(c)
    return n


def plain_leaf(column, op, literals, type_):
    """PLAIN leaves carry the literal's bit pattern in the 64-bit constant slots."""
    v = np.ascontiguousarray(np.atleast_1d(literals), dtype=NP_TYPES[type_])
    raw = np.zeros(len(v), dtype=np.uint64)
    raw.view(np.uint8).reshape(len(v), 8)[:, :v.itemsize] = v.view(np.uint8).reshape(len(v), -1)
    return leaf(column, op, raw)


def and_node():
    n = Node()
    n.kind = NODE_AND
    return n


def or_node():
    n = Node()
    n.kind = NODE_OR
    return n


def fle_column(enc, bw):
    c = Column()
    c.encoding, c.bit_width, c.type, c.d_data = COL_FLE, bw, 0, enc.data_ptr()
    return c


def nullable_fle_column(def_levels, def_bw, max_def, enc, bw, n_data_rows):
    """An OPTIONAL FLE column: definition levels of all rows + the data blocks of the non-NULL rows."""
    c = Column()
    c.encoding, c.bit_width, c.type, c.d_data = COL_FLE, bw, 0, enc.data_ptr()
    c.max_def_level, c.d_def_levels, c.def_bit_width, c.n_data_rows = max_def, def_levels.data_ptr(), def_bw, n_data_rows
    return c


def plain_column(page, type_):
    c = Column()
    c.encoding, c.bit_width, c.type, c.d_data = COL_PLAIN, 0, type_, page.data_ptr()
    return c


PROGRAM_AUTO, PROGRAM_PER_OPERAND, PROGRAM_ONE_PASS, PROGRAM_ONE_LAUNCH = range(4)


def set_program_strategy(strategy):
    _ck(lib().ips_set_program_strategy(int(strategy)))


def program_workspace_bytes(nodes, cols, n_rows):
    arr_n = (Node * len(nodes))(*nodes)
    arr_c = (Column * len(cols))(*cols)
    return int(lib().ips_program_workspace_bytes(arr_n, len(nodes), arr_c, len(cols), n_rows))


def eval_program(nodes, cols, n_rows, bitmap=None, device=None, stream=None, workspace=None):
    """workspace: uint8 tensor of program_workspace_bytes() bytes (allocated here when the tree
    needs one and none is given; pass it to keep the call allocation-free, e.g. under capture)."""
    arr_n = (Node * len(nodes))(*nodes)
    arr_c = (Column * len(cols))(*cols)
    device = device or (bitmap.device if bitmap is not None else torch.device("cuda"))
    if bitmap is None:
        bitmap = torch.empty(max(_words(n_rows), 2), dtype=torch.int64, device=device)
    if workspace is None:
        need = int(lib().ips_program_workspace_bytes(arr_n, len(nodes), arr_c, len(cols), n_rows))
        if need:
            workspace = torch.empty(need, dtype=torch.uint8, device=device)
    _ck(lib().ips_eval_program(arr_n, len(nodes), arr_c, len(cols), C.c_int64(n_rows),
                               _ptr(bitmap), _ptr(workspace), _stream(stream)))
    return bitmap[:_words(n_rows)]


# ---- IN lists of any length ---------------------------------------------------------------------
class InSet:
    """ips_inset: InSet(constants) for FLE values / codes, InSet(literals, dict_=d) for a dictionary column."""

    def __init__(self, values, dict_=None):
        self.h = C.c_void_p(0)
        if dict_ is None:
            v = np.ascontiguousarray(np.atleast_1d(values), dtype=np.uint64)
            _ck(lib().ips_inset_open(v.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_int64(len(v)), C.byref(self.h)))
        else:
            v = np.ascontiguousarray(np.atleast_1d(values), dtype=NP_TYPES[dict_.type])
            _ck(lib().ips_dict_inset_open(dict_.h, v.ctypes.data_as(C.c_void_p), C.c_int64(len(v)), C.byref(self.h)))
        lib().ips_inset_size.restype = C.c_int64
        lib().ips_inset_size.argtypes = [C.c_void_p]
        self.size = int(lib().ips_inset_size(self.h))

    def close(self):
        if self.h:
            lib().ips_inset_close(self.h)
            self.h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fle_pred_inset(enc, n_rows, bw, inset, bitmap=None, stream=None):
    if bitmap is None:
        bitmap = torch.empty(max(_words(n_rows), 2), dtype=torch.int64, device=enc.device)
    _ck(lib().ips_fle_pred_inset(_ptr(enc), C.c_int64(n_rows), bw, inset.h, _ptr(bitmap), _stream(stream)))
    return bitmap[:_words(n_rows)]


def fle_scan_inset(enc, n_rows, bw, inset, outputs=None, stream=None):
    bitmap, bvals, counts = outputs or alloc_scan_outputs(n_rows, enc.device)
    _ck(lib().ips_fle_scan_inset(_ptr(enc), C.c_int64(n_rows), bw, inset.h, _ptr(bitmap), _ptr(bvals), _ptr(counts),
                                 _stream(stream)))
    return bitmap[:_words(n_rows)], bvals, counts[:n_batches(n_rows)]


def dict_scan_inset(dict_, codes_enc, n_rows, bw, inset, stream=None):
    bitmap, bvals, counts = alloc_scan_outputs(n_rows, codes_enc.device, TORCH_SLOT[dict_.type])
    _ck(lib().ips_dict_scan_inset(dict_.h, _ptr(codes_enc), C.c_int64(n_rows), bw, inset.h, _ptr(bitmap), _ptr(bvals),
                                  _ptr(counts), _stream(stream)))
    return bitmap[:_words(n_rows)], bvals, counts[:n_batches(n_rows)]


def inset_leaf(column, inset):
    n = Node()
    n.kind, n.column, n.op, n.n_consts = NODE_LEAF, column, OP_IN, 0
    n.inset = inset.h
    return n


# ---- column chunks as lists of pages ----------------------------------------------------------
class ChunkPage(C.Structure):
    _fields_ = [("d_data", C.c_void_p), ("n_rows", C.c_int64), ("bit_width", C.c_int32),
                ("reserved", C.c_int32), ("d_def_levels", C.c_void_p), ("n_data_rows", C.c_int64)]


class Chunk:
    """ips_chunk: pages = [(data tensor, n_rows, bit_width[, def_levels tensor, n_data_rows])];
    the tensors are kept alive by the object."""

    def __init__(self, pages, encoding=COL_FLE, type_=T_INT32, max_def_level=0):
        self.keep = pages
        arr = (ChunkPage * max(len(pages), 1))()
        for i, pg in enumerate(pages):
            arr[i].d_data = pg[0].data_ptr() if pg[0] is not None else 0
            arr[i].n_rows = pg[1]
            arr[i].bit_width = pg[2]
            if len(pg) > 3:
                arr[i].d_def_levels = pg[3].data_ptr()
                arr[i].n_data_rows = pg[4]
        self.h = C.c_void_p()
        _ck(lib().ips_chunk_open(arr, len(pages), encoding, type_, max_def_level, C.byref(self.h)))
        L = lib()
        L.ips_chunk_num_rows.restype = C.c_int64
        L.ips_chunk_num_batches.restype = C.c_int64
        L.ips_chunk_num_rows.argtypes = [C.c_void_p]
        L.ips_chunk_num_batches.argtypes = [C.c_void_p]
        self.n_rows = int(L.ips_chunk_num_rows(self.h))
        self.n_batches = int(L.ips_chunk_num_batches(self.h))
        self.type = type_
        self.encoding = encoding
        self.device = next((pg[0].device for pg in pages if pg[0] is not None), torch.device("cuda"))

    def close(self):
        if self.h:
            lib().ips_chunk_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def alloc_outputs(self, value_dtype=torch.int32):
        dev = self.device
        bitmap = torch.empty(max(_words(self.n_rows), 2), dtype=torch.int64, device=dev)
        bvals = torch.empty(max(self.n_batches, 1) * BATCH_ROWS, dtype=value_dtype, device=dev)
        counts = torch.empty(max(self.n_batches, 1), dtype=torch.int32, device=dev)
        return bitmap, bvals, counts

    def _ret(self, outs):
        return outs[0][:_words(self.n_rows)], outs[1], outs[2][:self.n_batches]

    def fle_scan(self, op, values, outputs=None, stream=None):
        outs = outputs or self.alloc_outputs()
        keep, cs, k = _consts(values)
        _ck(lib().ips_chunk_fle_scan(self.h, op, cs, k, _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]), _stream(stream)))
        return self._ret(outs)

    def dict_scan(self, dict_, op, literals, outputs=None, stream=None):
        outs = outputs or self.alloc_outputs(TORCH_SLOT[dict_.type])
        keep, lp, k = dict_._lits(literals)
        _ck(lib().ips_chunk_dict_scan(self.h, dict_.h, op, lp, k, _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]),
                                      _stream(stream)))
        return self._ret(outs)

    def plain_scan(self, op, literals, op2=None, literal2=None, semantics=SEM_SQL, outputs=None, stream=None):
        stride = int(lib().ips_plain_stride(self.type))
        outs = outputs or self.alloc_outputs(torch.int32 if stride == 4 else torch.int64)
        v = np.ascontiguousarray(np.atleast_1d(literals), dtype=NP_TYPES[self.type])
        v2 = None if literal2 is None else np.ascontiguousarray(np.atleast_1d(literal2), dtype=NP_TYPES[self.type])
        _ck(lib().ips_chunk_plain_scan(self.h, op, v.ctypes.data_as(C.c_void_p), len(v), 0 if op2 is None else op2,
                                       None if v2 is None else v2.ctypes.data_as(C.c_void_p), semantics,
                                       _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]), _stream(stream)))
        return self._ret(outs)

    def select(self, bitmap, dict_=None, outputs=None, stream=None):
        if self.encoding == COL_PLAIN:
            dt = torch.int32 if int(lib().ips_plain_stride(self.type)) == 4 else torch.int64
        else:
            dt = TORCH_SLOT[dict_.type] if dict_ else torch.int32
        outs = outputs or self.alloc_outputs(dt)
        _ck(lib().ips_chunk_select(self.h, dict_.h if dict_ else None, _ptr(bitmap), _ptr(outs[1]), _ptr(outs[2]),
                                   _stream(stream)))
        return outs[1], outs[2][:self.n_batches]

    def compact(self, bvals, counts, stream=None):
        return batches_compact(bvals, counts, self.n_batches * BATCH_ROWS, stream=stream)

    def select_nullable(self, selection, dict_=None, stream=None):
        """ips_chunk_select_nullable on an OPTIONAL chunk -> (dense values of the selected NOT-NULL rows,
        NOT-NULL flag words of the selected rows, n_selected, n_values, bad_index); synchronises for the counts."""
        L = lib()
        L.ips_chunk_select_nullable_workspace_bytes.restype = C.c_size_t
        L.ips_chunk_select_nullable_workspace_bytes.argtypes = [C.c_void_p]
        need = int(L.ips_chunk_select_nullable_workspace_bytes(self.h))
        ws = torch.empty(need + 16, dtype=torch.uint8, device=self.device)
        dt = TORCH_SLOT[dict_.type] if dict_ else torch.int32
        dense = torch.empty(max(self.n_rows, 4), dtype=dt, device=self.device)
        flags = torch.empty(max(_words(self.n_rows), 2), dtype=torch.int64, device=self.device)
        counts = torch.empty(3, dtype=torch.int64, device=self.device)
        _ck(L.ips_chunk_select_nullable(self.h, dict_.h if dict_ else None, _ptr(selection), _ptr(dense), _ptr(flags),
                                        _ptr(counts), _ptr(ws), _stream(stream)))
        c = counts.cpu().tolist()
        return dense[:c[1]], flags[:_words(c[0])], int(c[0]), int(c[1]), int(c[2])


def eval_program_chunks(nodes, chunks, bitmap=None, workspace=None, stream=None):
    """ips_eval_program_chunks: leaf.column indexes chunks."""
    arr_n = (Node * len(nodes))(*nodes)
    arr_c = (C.c_void_p * len(chunks))(*[c.h for c in chunks])
    n_rows = chunks[0].n_rows
    dev = chunks[0].device
    L = lib()
    L.ips_chunk_program_workspace_bytes.restype = C.c_size_t
    need = int(L.ips_chunk_program_workspace_bytes(arr_n, len(nodes), arr_c, len(chunks)))
    if workspace is None and need:
        workspace = torch.empty(need + 16, dtype=torch.uint8, device=dev)
    if bitmap is None:
        bitmap = torch.empty(max(_words(n_rows), 2), dtype=torch.int64, device=dev)
    _ck(L.ips_eval_program_chunks(arr_n, len(nodes), arr_c, len(chunks), _ptr(bitmap), _ptr(workspace),
                                  _stream(stream)))
    return bitmap[:_words(n_rows)]


# ---- multi-GPU exchange (native RCCL path; bench.py uses torch.distributed) -------------------
COMM_ID_BYTES = 128


def comm_unique_id():
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _ck(lib().ips_comm_unique_id(buf, COMM_ID_BYTES))
    return buf.raw


class Comm:
    def __init__(self, id_bytes, nranks, rank):
        self.h = C.c_void_p(0)
        self.nranks = nranks
        _ck(lib().ips_comm_init(C.c_char_p(id_bytes), nranks, rank, C.byref(self.h)))

    def allgather_bitmap(self, local_words, stream=None):
        out = torch.empty(local_words.numel() * self.nranks, dtype=local_words.dtype,
                          device=local_words.device)
        _ck(lib().ips_allgather_bitmap(self.h, _ptr(local_words), C.c_int64(local_words.numel()),
                                       _ptr(out), _stream(stream)))
        return out

    def fle_scan_allgather(self, enc, n_rows, bw, op, values, n_chunks, local_bitmap, bvals, counts,
                           all_bitmap, stream=None):
        """One step of the sharded scan: chunked scan + overlapped all-gather (one C call)."""
        keep, p, k = _consts(values)
        _ck(lib().ips_fle_scan_allgather(self.h, _ptr(enc), C.c_int64(n_rows), bw, op, p, k, n_chunks,
                                         _ptr(local_bitmap), _ptr(bvals), _ptr(counts), _ptr(all_bitmap),
                                         _stream(stream)))

    def join(self, stream=None):
        _ck(lib().ips_comm_join(self.h, _stream(stream)))

    def eval_program_chunks_allgather(self, nodes, chunks, local_bitmap, full_bitmap, workspace=None, stream=None):
        """ips_eval_program_chunks_allgather: one call per step of the sharded predicate tree."""
        arr_n = (Node * len(nodes))(*nodes)
        arr_c = (C.c_void_p * len(chunks))(*[c.h for c in chunks])
        _ck(lib().ips_eval_program_chunks_allgather(self.h, arr_n, len(nodes), arr_c, len(chunks), _ptr(local_bitmap),
                                                    _ptr(full_bitmap), _ptr(workspace), _stream(stream)))

    def check(self):
        _ck(lib().ips_comm_check(self.h))

    def close(self):
        if self.h:
            lib().ips_comm_destroy(self.h)
            self.h = C.c_void_p(0)


# ---- synthetic ------------------------------------------------------------------------------
def synth_u32(seed, n, bit_width, device=None, stream=None):
    out = torch.empty(max(n, 16), dtype=torch.int32, device=device or torch.device("cuda"))
    mask = (1 << bit_width) - 1
    _ck(lib().ips_synth_splitmix_u32(C.c_uint64(seed), C.c_int64(n), C.c_uint32(mask), _ptr(out),
                                     _stream(stream)))
    return out[:n]

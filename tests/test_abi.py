"""CPU tests of the drop-in boundary: libips_hip.so loads without a GPU and exports exactly the
symbols include/ips.h declares; argument validation works before any device is touched."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "ips.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ips_[a-z0-9_]+)\s*\(", text)))


def test_header_matches_binding(ips):
    assert header_functions() == sorted(ips.capi.SYMBOLS)


def test_library_exports_every_symbol(ips):
    lib = ips.capi.lib()
    for name in header_functions():
        assert hasattr(lib, name), name
    assert lib.ips_version() == 301


def test_pure_host_entry_points(ips):
    lib = ips.capi.lib()
    # FleEncoder::Flush lengths asserted by fle-test.cc:219,224,238
    assert lib.ips_fle_encoded_bytes(100, 1) == 16
    assert lib.ips_fle_encoded_bytes(100, 2) == 32
    assert lib.ips_fle_encoded_bytes(0, 7) == 0
    assert lib.ips_fle_encoded_bytes(1 << 20, 32) == 4 << 20
    # DictEncoderBase::bit_width, dict-encoding.h:76-80
    assert [lib.ips_dict_bit_width(d) for d in (0, 1, 2, 3, 256, 257, 40000)] == [0, 1, 1, 2, 8, 9, 16]
    # ParquetPlainEncoder::ByteSize, parquet-common.h:92-117
    assert [lib.ips_plain_stride(t) for t in range(6)] == [4, 4, 4, 8, 4, 8]


def test_argument_validation_without_device(ips):
    """Invalid bit width / op / alignment are errors, never UB (SURVEY 8b 'Errors')."""
    capi = ips.capi
    lib = capi.lib()
    consts = (C.c_uint64 * 1)(5)
    null = C.c_void_p(0)
    fake = C.c_void_p(0x1000)         # never dereferenced: validation fails first
    assert lib.ips_fle_pred(fake, C.c_int64(64), 0, 1, consts, 1, fake, null) == 1
    assert b"bit_width" in lib.ips_last_error()
    assert lib.ips_fle_pred(fake, C.c_int64(64), 33, 1, consts, 1, fake, null) == 1
    assert lib.ips_fle_pred(fake, C.c_int64(64), 8, 9, consts, 1, fake, null) == 1
    assert lib.ips_fle_pred(C.c_void_p(0x1004), C.c_int64(64), 8, 1, consts, 1, fake, null) == 1
    assert b"aligned" in lib.ips_last_error()
    assert lib.ips_fle_pred(fake, C.c_int64(-1), 8, 1, consts, 1, fake, null) == 1
    assert lib.ips_fle_decode(fake, C.c_int64(64), 9, fake, 1, null) == 1
    assert lib.ips_plain_pred(fake, C.c_int64(64), 9, 1, consts, 1, 1, fake, null) == 1
    # IN on a PLAIN page has no reference behaviour (empty body, parquet-common.h:252-255)
    assert lib.ips_plain_pred(fake, C.c_int64(64), 2, 5, consts, 1, 0, fake, null) == 2
    # n_rows == 0 is a no-op
    assert lib.ips_fle_pred(null, C.c_int64(0), 8, 1, consts, 1, null, null) == 0

"""CPU test of the page container the host facade parses (thrift compact-protocol PageHeader,
GZIP, Snappy): tests/host_page_header_test.cpp, built with g++ + zlib and ASan/UBSan, no GPU.
The Snappy restatement is cross-checked both ways against libsnappy as bundled in pyarrow."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "host_page_header_test")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wextra", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-o", exe,
                           os.path.join(ROOT, "tests", "host_page_header_test.cpp"), "-lz"])
    return exe


def test_page_header_reader_writer_and_codecs(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failed" in r.stdout


def test_snappy_against_libsnappy(tmp_path):
    """Pages compressed by libsnappy (pyarrow's bundled copy) decompress with the facade's codec,
    and the facade's compressor's output decompresses with libsnappy: FLE blocks, PLAIN int64,
    text, runs, noise, a 3 MiB page (copies with 2-byte offsets across 64-KiB fragments)."""
    pa = pytest.importorskip("pyarrow")
    if not pa.Codec.is_available("snappy"):
        pytest.skip("pyarrow built without snappy")
    codec = pa.Codec("snappy")
    exe = _build(tmp_path)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    rng = np.random.default_rng(11)
    pages = {
        "empty": b"",
        "one": b"x",
        "text": b"the quick brown fox jumps over the lazy dog. " * 500,
        "runs": bytes(100000),
        "noise": rng.integers(0, 256, 200000, dtype=np.uint8).tobytes(),
        "plain_i64": (rng.integers(0, 300, 50000) * 1000003).astype(np.int64).tobytes(),
        "fle_planes": np.repeat(rng.integers(0, 2 ** 63, 4096, dtype=np.uint64), 3).tobytes(),
        "big": (rng.integers(0, 50, 3 << 20, dtype=np.uint8) * 5).tobytes(),
    }
    for name, raw in pages.items():
        src, comp, back = tmp_path / (name + ".raw"), tmp_path / (name + ".sz"), tmp_path / (name + ".out")
        # libsnappy -> ours
        comp.write_bytes(codec.compress(raw, asbytes=True))
        r = subprocess.run([exe, "snappy-d", str(comp), str(back), str(len(raw))], env=env, capture_output=True, timeout=120)
        assert r.returncode == 0, (name, r.stderr)
        assert back.read_bytes() == raw, name
        # ours -> libsnappy
        src.write_bytes(raw)
        r = subprocess.run([exe, "snappy-c", str(src), str(comp)], env=env, capture_output=True, timeout=120)
        assert r.returncode == 0, (name, r.stderr)
        ours = comp.read_bytes()
        assert codec.decompress(ours, decompressed_size=len(raw), asbytes=True) == raw, name
        if name in ("text", "runs"):
            assert len(ours) < len(raw) // 8


def test_column_chunk_walk_and_fuzz(tmp_path):
    """tests/host_column_chunk_test.cpp: the host-only column-chunk walk + data-page framing on valid
    chunks (3 codecs x PLAIN / dictionary x REQUIRED / OPTIONAL) and 18 000 corrupted ones, ASan + UBSan."""
    exe = str(tmp_path / "host_column_chunk_test")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wextra", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-o", exe,
                           os.path.join(ROOT, "tests", "host_column_chunk_test.cpp"), "-lz"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 failed" in r.stdout

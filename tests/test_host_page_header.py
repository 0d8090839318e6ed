"""CPU test of the page container the host facade parses (thrift compact-protocol PageHeader,
GZIP): tests/host_page_header_test.cpp, built with g++ + zlib and ASan/UBSan, no GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_page_header_reader_writer_and_codecs(tmp_path):
    exe = str(tmp_path / "host_page_header_test")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wextra", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-o", exe,
                           os.path.join(ROOT, "tests", "host_page_header_test.cpp"), "-lz"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failed" in r.stdout

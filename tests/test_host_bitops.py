"""CPU test: the per-lane bit arithmetic shared with the HIP kernels (csrc/ips_bitops.h) compiled
with g++ and checked against the FLE layout definition for every bit width."""
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))


def test_bitops_on_host():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "host_bitops_test")
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", exe,
                               os.path.join(HERE, "host_bitops_test.cpp")])
        r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr

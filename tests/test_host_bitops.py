"""CPU test: the per-lane bit arithmetic shared with the HIP kernels (csrc/ips_bitops.h) compiled
with g++ and checked against the FLE layout definition for every bit width."""
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))


def test_bitops_on_host():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "host_bitops_test")
        # AddressSanitizer + UBSan: the GPU pool has no sanitizers, the shared arithmetic gets them here
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined",
                               "-fno-sanitize-recover=undefined", "-o", exe,
                               os.path.join(HERE, "host_bitops_test.cpp")])
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
        env.pop("LD_PRELOAD", None)
        r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout + r.stderr

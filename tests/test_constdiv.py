"""Proof obligation of the arithmetic contract: the 3-instruction constant divides the
kernels use for /9 (3x3 window mean) and /3 (channel mean) equal the IEEE quotient for
EVERY finite float.  Exhaustive (2 x 2^32 values, ~45 s on 8 cores)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_constant_divides_are_exact_for_all_floats():
    src = os.path.join(ROOT, "oracle", "check_constdiv.c")
    exe = os.path.join(ROOT, "oracle", "_build", "check_constdiv")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    fma = ["-mfma"] if "fma" in open("/proc/cpuinfo").read() else []
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", *fma, "-o", exe, src, "-lm"])
    out = subprocess.run([exe, "9", "3"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout
    assert out.stdout.count("mismatches=0") == 2, out.stdout

"""GPU (-m gpu): a reference user's C++ program (tests/cpp/facade_test.cpp) that factorizes with the real
hif::HIF on the host and then routes solve / solve(trans) / solve_mrhs / hifir / mmultiply / GMRES through
the header-only facade include/hifir_amd.hpp, comparing every call with what the reference returns.
The binary is built by oracle/Makefile into oracle/_ref/ (only where the reference exists) and travels
to the GPU box prebuilt, like the compiled reference itself."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "facade_test")


@pytest.mark.skipif(not os.path.exists(BIN), reason="facade_test not built (needs the reference headers)")
def test_cpp_facade_matches_reference_call_for_call():
    out = subprocess.run([BIN], capture_output=True, text=True, timeout=600)
    print(out.stdout)
    print(out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "FACADE TEST OK" in out.stdout
    assert out.stdout.count(" ok") >= 9

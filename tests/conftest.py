import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "perf: wall-clock expectations on a GPU box (run with -m perf; not part of -m gpu)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref/libhifref.so (the compiled reference)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    # building the checker is not using it; the C restatement compiles in a second
    from oracle import orc

    orc.lib()

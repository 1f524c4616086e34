"""Wall-clock expectations (-m perf, never part of -m gpu: a noisy box must not turn the parity suite red)."""
import time

import numpy as np
import pytest

import hifir_amd
from util import load_hier

pytestmark = pytest.mark.perf


def test_rotating_buffers_cost_no_recapture():
    # a Krylov solver hands over a different (B, X) pair every call: the captured graph is keyed by shape and reads
    # the pointers from a device slot, so rotating 12 pairs must cost what reusing one pair costs
    torch = pytest.importorskip("torch")
    if hifir_amd.lib().hifamd_device_count() == 0:
        pytest.skip("needs a GPU")
    levels, d = load_hier("p2d_64_deep")
    M = hifir_amd.HIF.from_levels(levels, max_nrhs=16)
    n = len(d["b"])
    rng = np.random.default_rng(41)
    Bs = [torch.from_numpy(rng.uniform(-1, 1, size=(n, 16))).cuda() for _ in range(12)]
    Xs = [torch.empty_like(b) for b in Bs]
    M.solve_mrhs(Bs[0], Xs[0])
    M.sync()

    def wall(pairs):
        t0 = time.perf_counter()
        for b, x in pairs:
            M.solve_mrhs(b, x)
        M.sync()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / len(pairs)

    same = min(wall([(Bs[0], Xs[0])] * len(Bs)) for _ in range(3))
    rot = min(wall(list(zip(Bs, Xs))) for _ in range(3))
    assert rot <= 1.5 * same + 1e-3, (rot, same)  # a re-capture per call would cost tens of milliseconds

"""The N > 1 flow of bench.py on ONE GPU (driver-run suite): two ranks as a fresh child process group -- the parent test
process makes no launch of its own for it --, gloo for the barriers, both ranks on device 0.  Checks what the 8-GPU run
relies on: rank 0 factorizes and writes the hierarchy file with its analysis trailer, rank 1 loads it and ADOPTS the
analysis, every rank applies its own column block, the strong split and the end-of-batch gather are reported."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_gloo_one_gpu(tmp_path):
    env = dict(os.environ)
    env["TMPDIR"] = str(tmp_path)  # (the hierarchy file of this run only)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--nx", "300", "--steps", "3",
           "--secondary", "0", "--extras", "0", "--cpu-seconds", "0.5"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert line["value"] > 0 and line["ms_per_step"] > 0
    assert "strong_scaling" in line and "gather_ms" in line
    by_rank = line["config"]["analysis_levels_from_file_by_rank"]
    assert len(by_rank) == 2 and by_rank[0] == 0  # rank 0 analyzed what it factorized ...
    assert by_rank[1] == line["config"]["levels"]  # ... rank 1 adopted every level's analysis from the file

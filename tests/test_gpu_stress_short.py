"""Short runs of the round-5 randomised parity stresses (scripts/stress_*.py; the long runs: profiles/r05/stress_end_of_round.txt) inside the GPU
suite: random shapes, types, channel counts and chunkings, each script against its own checker (the oracle, the universal kernel, one unsharded /
unchained / uncaptured filter).  Every script runs as a process of its own (it opens rings, captures graphs, sets switches per case)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,args", [
    ("stress_blocks.py", ["--cases", "150", "--seconds", "40", "--seed", "201"]),        # L > 512 in period blocks
    ("stress_sharded.py", ["--cases", "150", "--seconds", "40", "--seed", "202"]),       # mrhip_sharded_*
    ("stress_cascade.py", ["--cases", "150", "--seconds", "40", "--seed", "203"]),       # chained device-planned calls
    ("stress_graph.py", ["--cases", "150", "--seconds", "40", "--seed", "204"]),         # captured calls, history written in place
    ("stress_multi.py", ["--cases", "80", "--seconds", "40", "--seed", "205"]),          # mrhip_filt_device_multi
])
def test_randomised_stress_short(script, args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script)] + args, capture_output=True, text=True, timeout=400)
    assert p.returncode == 0, (script, p.stdout[-800:], p.stderr[-800:])
    assert "mismatches 0" in p.stdout, p.stdout[-400:]

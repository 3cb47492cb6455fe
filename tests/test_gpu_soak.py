"""Short, seeded runs of the randomised soak scripts (tests/soak/) inside the GPU suite: each script checks every
result against the oracle and prints "<n> mismatches" on its last line."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("script,args", [
    ("fuzz_convs.py", ["20261", "60"]),
    ("fuzz_small_channel_convs.py", ["20262", "40"]),
    ("fuzz_graphs.py", ["20263", "40"]),
    ("fuzz_graphs_f32.py", ["20264", "25"]),
    ("fuzz_tail.py", ["20265", "60"]),
    ("fuzz_api_states.py", ["20266", "60"]),
])
def test_soak_script(script, args):
    out = subprocess.run([sys.executable, os.path.join(HERE, "soak", script)] + args, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    last = out.stdout.strip().splitlines()[-1]
    m = re.search(r"(\d+) mismatches", last)
    assert m and int(m.group(1)) == 0, out.stdout[-2000:]

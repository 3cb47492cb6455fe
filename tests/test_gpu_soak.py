"""Short, seeded runs of the randomised soak scripts (tests/soak/) inside the GPU suite: each script checks every
result against the oracle and prints "<n> mismatches" on its last line."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("script,args,env", [
    ("fuzz_convs.py", ["20261", "60"], {}),
    ("fuzz_convs.py", ["20267", "40"], {"FUZZ_IC": "16"}),  # 16 input channels on maps around multiples of 16: the patch-staged kernel's round-6 shapes
    ("fuzz_small_channel_convs.py", ["20262", "40"], {}),
    ("fuzz_graphs.py", ["20263", "40"], {}),
    ("fuzz_graphs.py", ["20268", "40"], {"FUZZ_NCHW": "1"}),  # every graph NCHW-tagged: nhwc_internal, the byte-wise layers on the internal layout, virtual_concat_q
    ("fuzz_graphs_f32.py", ["20264", "25"], {}),
    ("fuzz_vcat_f32.py", ["20269", "25"], {}),  # float concats read through a view of their last input (virtual_concat_f32), modes 3 / 4
    ("fuzz_tail.py", ["20265", "60"], {}),
    ("fuzz_api_states.py", ["20266", "60"], {}),
], ids=lambda v: v if isinstance(v, str) else ("-".join(v) if isinstance(v, list) else "+".join(sorted(v)) or "default"))
def test_soak_script(script, args, env):
    out = subprocess.run([sys.executable, os.path.join(HERE, "soak", script)] + args, capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, **env))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    last = out.stdout.strip().splitlines()[-1]
    m = re.search(r"(\d+) mismatches", last)
    assert m and int(m.group(1)) == 0, out.stdout[-2000:]

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionfinish(session, exitstatus):
    """-m gpu runs: dump the error statistics every parity comparison recorded (tests/gpu_util.py) next to the other
    GPU-side outputs, so that the tolerances in the tests can be read against what was measured."""
    try:
        from tests import gpu_util
    except Exception:
        return
    if not gpu_util.STATS:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_stats.json"), "w") as f:
        json.dump([dict(zip(("what", "elements", "frac_beyond_maxnorm", "max_err_over_max", "frac_beyond_per_element"), r))
                   for r in gpu_util.STATS], f, indent=0)

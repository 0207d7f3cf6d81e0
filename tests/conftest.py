import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REPORT = []


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test")


def pytest_sessionfinish(session, exitstatus):
    # numeric error report of the GPU parity tests (read back from gpurun_out/ after a gpurun call)
    if REPORT:
        out = os.path.join(ROOT, "gpurun_out")
        try:
            os.makedirs(out, exist_ok=True)
            with open(os.path.join(out, "parity_report.json"), "w") as f:
                json.dump(REPORT, f, indent=1)
        except OSError:
            pass


@pytest.fixture
def report():
    return REPORT

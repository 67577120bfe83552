import base64
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """On a host without an AMD GPU device node, tests marked `gpu` are skipped rather than left to fail with
    TD_E_HIP (plain `pytest tests/` then shows real CPU-side regressions only)."""
    if os.path.exists("/dev/kfd"):
        return
    skip = pytest.mark.skip(reason="needs a GPU: /dev/kfd is absent on this host")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_sessionstart(session):
    """Built files stay out of history: bring libtagdig.so up to date with its sources (a no-op when it
    is; hipcc cross-compiles for gfx950 without a GPU) so that the C-ABI tests load what the tree says."""
    import shutil
    import subprocess
    if shutil.which("make") and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tagdigger_amd", "csrc")])


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


def case_payload(case):
    return base64.b64decode(case["fastq_b64"])


def write_case_file(case, directory):
    """Materialise a golden case's FASTQ under its recorded file name."""
    path = os.path.join(str(directory), case["filename"])
    with open(path, "wb") as fh:
        fh.write(case_payload(case))
    return path


@pytest.fixture(scope="session")
def golden_cases():
    return load_golden("hotpath_cases.json") + load_golden("hotpath_random.json")

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")


def pytest_collection_modifyitems(config, items):
    """`gpu`-marked tests need the built library AND a HIP device: without
    them they are skipped, not failed (a plain `pytest tests/` on the build host)."""
    reason = None
    try:
        import nxsearch_amd
        if nxsearch_amd.lib().nxsgpu_device_count() <= 0:
            reason = "no HIP device"
    except (ImportError, OSError) as e:
        reason = "libnxsearch_gpu.so not built: %s" % e
    if reason:
        skip = pytest.mark.skip(reason=reason)
        for it in items:
            if "gpu" in it.keywords:
                it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")) as f:
        return json.load(f)

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "reference: needs /root/reference (live Python oracle)")


def pytest_collection_modifyitems(config, items):
    from ref_harness import reference_available

    if not reference_available():
        skip = pytest.mark.skip(reason="/root/reference not present on this machine")
        for item in items:
            if "reference" in item.keywords:
                item.add_marker(skip)

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# aie_create starts a background compile (hiprtc) of specialised kernels for every configuration outside the build's
# instance families -- the product's default.  This suite creates ~150 such configurations for a few launches each: the
# jobs would only queue up behind one another.  Off here; test_background_specialisation_swaps_in_without_a_call turns it
# on for itself, the aie_specialize tests do not need it (an explicit request compiles on the spot).
os.environ.setdefault("AIE_JIT_AUTO", "0")


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

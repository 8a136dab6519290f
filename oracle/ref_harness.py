"""Test infrastructure ONLY: import the *unmodified* reference Foundation from
/root/reference in-process, so it can (a) validate the C restatement in
oracle/aie_oracle.c and (b) generate the golden vectors under tests/golden/.

Nothing in the product path (ai-economist_amd/, bench.py's GPU leg) may import this
module.  /root/reference does not exist on the GPU box, so everything here is guarded
by `reference_available()`.

Shims (none touch arithmetic):
  * lz4 / Crypto / GPUtil are imported at module top level by the reference
    (ai_economist/foundation/utils.py:12-13, components/covid19_components.py:9,
    scenarios/covid19/covid19_env.py:11) but are not installed -> stub modules.
  * NumPy >= 1.24 has no np.int (layout_from_file.py:212-213) -> np.int = int.
  * COVID only: verify_activation_code() blocks on input() (covid19_env.py:114).
"""
import os
import sys
import types

import numpy as np

# The live tree where it exists (build container); otherwise oracle/_ref/: the same package byte-compiled from
# /root/reference by oracle/make_ref.py (git-ignored, travels to the GPU box with the tree).
_REF_BUILT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")
REFERENCE_ROOT = os.environ.get("AIE_REFERENCE_ROOT") or (
    "/root/reference" if os.path.isdir("/root/reference/ai_economist/foundation") else _REF_BUILT)


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "ai_economist", "foundation"))


def reference_is_live_tree():
    """False when only the byte-compiled copy (oracle/_ref) is available."""
    return reference_available() and os.path.abspath(REFERENCE_ROOT) != os.path.abspath(_REF_BUILT)


_foundation = None


def load_reference_foundation():
    """Returns the reference `ai_economist.foundation` module (cached)."""
    global _foundation
    if _foundation is not None:
        return _foundation
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)

    if not hasattr(np, "int"):
        np.int = int  # noqa: NPY001  (shim for layout_from_file.py:212-213)

    def _stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules.setdefault(name, m)
        return sys.modules[name]

    lz4 = _stub("lz4")
    frame = _stub("lz4.frame", compress=lambda b: b, decompress=lambda b: b)
    lz4.frame = frame
    crypto = _stub("Crypto")
    pk = _stub("Crypto.PublicKey")
    rsa = _stub("Crypto.PublicKey.RSA", importKey=lambda *a, **k: None)
    crypto.PublicKey = pk
    pk.RSA = rsa
    _stub("GPUtil", getAvailable=lambda *a, **k: [])

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import ai_economist.foundation as foundation  # noqa: E402

    mod = sys.modules.get("ai_economist.foundation.scenarios.covid19.covid19_env")
    if mod is not None:
        mod.verify_activation_code = lambda: None
    _foundation = foundation
    return foundation

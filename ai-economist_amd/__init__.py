"""ai-economist_amd: MI355X-native batched Foundation env.step().

The directory is named `ai-economist_amd`; import it as `ai_economist_amd` (the shim
`ai_economist_amd.py` at the repository root maps the two).
"""
from . import foundation  # noqa: F401

__all__ = ["foundation"]

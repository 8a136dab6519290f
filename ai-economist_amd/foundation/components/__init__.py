"""Component registry + the gather-trade-build components.

Each class keeps the reference component's registry name, kwargs, defaults and
constructor validation, and knows how to write itself into the C-ABI config
(`aie_config`, include/aie.h).  The dynamics themselves run in the HIP kernels
(ai-economist_amd/csrc/aie_kernels.hip), not here.
"""
from .base import BaseComponent, BatchedComponent, component_registry  # noqa: F401
from . import build, continuous_double_auction, covid19_components, move, redistribution, simple_labor  # noqa: F401

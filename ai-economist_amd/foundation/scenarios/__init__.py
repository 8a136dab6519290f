from ..base_env import BaseEnvironment, scenario_registry  # noqa: F401
from . import covid19, dynamic_layout, layout_from_file, one_step_economy  # noqa: F401

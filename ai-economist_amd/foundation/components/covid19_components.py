"""The three COVID-19 components (reference: F/components/covid19_components.py:32-663).
Registry names, kwargs, defaults and constructor checks are the reference's; the dynamics,
observations and masks of all three run inside the fused kernel
csrc/aie_kernels_covid.hip (aie_covid_step_kernel)."""
from datetime import datetime

from ... import _cabi
from .base import BaseComponent, component_registry


@component_registry.add
class ControlUSStateOpenCloseStatus(BaseComponent):
    """Sets the open/close stringency level of every US state (covid19_components.py:32-241)."""
    name = "ControlUSStateOpenCloseStatus"
    required_entities = []
    agent_subclasses = ["BasicMobileAgent"]
    comp_id = _cabi.COMP_COVID_CONTROL

    def __init__(self, *base_args, n_stringency_levels=10, action_cooldown_period=28, **base_kwargs):
        self.action_cooldown_period = action_cooldown_period
        super().__init__(*base_args, **base_kwargs)
        self.n_stringency_levels = int(n_stringency_levels)
        assert self.n_stringency_levels >= 2

    def get_n_actions(self, agent_cls_name):
        return self.n_stringency_levels if agent_cls_name == "BasicMobileAgent" else None

    def fill_config(self, cfg):
        cfg.covid.action_cooldown_period = int(self.action_cooldown_period)
        cfg.covid.num_stringency_levels = self.n_stringency_levels


@component_registry.add
class FederalGovernmentSubsidy(BaseComponent):
    """Direct payments from the federal government to the states (covid19_components.py:244-469)."""
    name = "FederalGovernmentSubsidy"
    required_entities = []
    agent_subclasses = ["BasicPlanner"]
    comp_id = _cabi.COMP_COVID_SUBSIDY

    def __init__(self, *base_args, subsidy_interval=90, num_subsidy_levels=20,
                 max_annual_subsidy_per_person=20000, **base_kwargs):
        self.subsidy_interval = int(subsidy_interval)
        assert self.subsidy_interval >= 1
        self.num_subsidy_levels = int(num_subsidy_levels)
        assert self.num_subsidy_levels >= 1
        self.max_annual_subsidy_per_person = float(max_annual_subsidy_per_person)
        assert self.max_annual_subsidy_per_person >= 0
        super().__init__(*base_args, **base_kwargs)

    def get_n_actions(self, agent_cls_name):
        return self.num_subsidy_levels if agent_cls_name == "BasicPlanner" else None

    def fill_config(self, cfg):
        cfg.covid.subsidy_interval = self.subsidy_interval
        cfg.covid.num_subsidy_levels = self.num_subsidy_levels


@component_registry.add
class VaccinationCampaign(BaseComponent):
    """Passive component delivering vaccines once the delivery date has passed
    (covid19_components.py:472-663)."""
    name = "VaccinationCampaign"
    required_entities = []
    agent_subclasses = ["BasicMobileAgent"]
    comp_id = _cabi.COMP_COVID_VACCINE

    def __init__(self, *base_args, daily_vaccines_per_million_people=4500, delivery_interval=1,
                 vaccine_delivery_start_date="2020-12-22", observe_rate=False, **base_kwargs):
        self.daily_vaccines_per_million_people = int(daily_vaccines_per_million_people)
        assert 0 <= self.daily_vaccines_per_million_people <= 1e6
        self.delivery_interval = int(delivery_interval)
        assert 1 <= self.delivery_interval <= 5000
        self.vaccine_delivery_start_date = datetime.strptime(vaccine_delivery_start_date, "%Y-%m-%d")
        self.observe_rate = bool(observe_rate)
        super().__init__(*base_args, **base_kwargs)

    def get_n_actions(self, agent_cls_name):
        return None  # passive

    def fill_config(self, cfg):
        cfg.covid.delivery_interval = self.delivery_interval

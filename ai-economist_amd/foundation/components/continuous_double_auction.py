"""`ContinuousDoubleAuction` (reference: F/components/continuous_double_auction.py:16-98,
411-431; dynamics -> cda_component_step / cda_match_orders in csrc/aie_kernels.hip)."""
from ... import _cabi
from .base import BaseComponent, component_registry


@component_registry.add
class ContinuousDoubleAuction(BaseComponent):
    name = "ContinuousDoubleAuction"
    component_type = "Trade"
    required_entities = ["Coin", "Labor"]
    agent_subclasses = ["BasicMobileAgent"]
    comp_id = _cabi.COMP_CDA

    def __init__(self, *base_args, max_bid_ask=10, order_labor=0.25, order_duration=50,
                 max_num_orders=None, **base_kwargs):
        super().__init__(*base_args, **base_kwargs)
        self.max_bid_ask = int(max_bid_ask)
        assert self.max_bid_ask >= 1
        self.price_floor = 0
        self.price_ceiling = int(max_bid_ask)
        self.order_duration = int(order_duration)
        assert self.order_duration >= 1
        self.max_num_orders = int(max_num_orders or self.order_duration)
        assert self.max_num_orders >= 1
        self.order_labor = max(float(order_labor), 0.0)
        self.commodities = ["Stone", "Wood"]  # sorted collectible resources

    def get_n_actions(self, agent_cls_name):
        if agent_cls_name == "BasicMobileAgent":
            trades = []
            for c in self.commodities:
                trades.append(("Buy_{}".format(c), 1 + self.max_bid_ask))
                trades.append(("Sell_{}".format(c), 1 + self.max_bid_ask))
            return trades
        return None

    def fill_config(self, cfg):
        cfg.cda_max_bid_ask = self.max_bid_ask
        cfg.cda_order_duration = self.order_duration
        cfg.cda_max_num_orders = self.max_num_orders
        cfg.cda_order_labor = self.order_labor

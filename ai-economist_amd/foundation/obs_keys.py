"""Key tables of the flat observation vectors.

The reference packs every scalar / 1-D observation of an agent into one float32 vector `flat`, concatenated in
SORTED-KEY order (F/base/base_env.py:561-612 `_build_packager` / `_package`); with `flatten_observations=False` it
hands the same values out under their own keys instead (the reference's own unit test builds its environment that
way, tests/test_env.py:40).  The kernels always write the packed vectors (csrc/aie_layout.h: fa_* / fp_* / fpa_*);
this module knows which slice of them each key is, so the unflattened form is a zero-copy view:

    flat_keys(env) -> {"a": [(key, offset, size, is_scalar)], "p": [...], "pa": [...]}

Key names: "<Component.name>-<key>" (base_env.py:644-673), "world-<key>" for the scenario's, "time".  The tables are
pinned against the live reference in tests/test_obs_keys.py.
"""

_CDA_AGENT = ["available_asks", "available_bids", "market_rate", "my_asks", "my_bids", "price_history"]
_CDA_PLANNER = ["full_asks", "full_bids", "market_rate", "price_history"]


def _component_keys(env):
    """(agent, planner, planner-per-agent) [(key, size, scalar)] of every component, unsorted."""
    n = env.n_agents
    a, p, pa = [], [], []
    for comp in env.components:
        name = comp.name
        if name == "Build":
            a += [("Build-build_payment", 1, True), ("Build-build_skill", 1, True)]
        elif name == "Gather":
            a += [("Gather-bonus_gather_prob", 1, True)]
        elif name == "ContinuousDoubleAuction":
            P = int(comp.max_bid_ask) + 1
            for res in ("Stone", "Wood"):
                for k in _CDA_AGENT:
                    a.append(("%s-%s-%s" % (name, k, res), 1 if k == "market_rate" else P, k == "market_rate"))
                for k in _CDA_PLANNER:
                    p.append(("%s-%s-%s" % (name, k, res), 1 if k == "market_rate" else P, k == "market_rate"))
        elif name == "PeriodicBracketTax":
            nb = int(comp.n_brackets)
            a += [(name + "-curr_rates", nb, False), (name + "-is_first_day", 1, True), (name + "-is_tax_day", 1, True),
                  (name + "-last_incomes", n, False), (name + "-marginal_rate", 1, True), (name + "-tax_phase", 1, True)]
            p += [(name + "-curr_rates", nb, False), (name + "-is_first_day", 1, True), (name + "-is_tax_day", 1, True),
                  (name + "-last_incomes", n, False), (name + "-tax_phase", 1, True)]
            pa += [(name + "-curr_marginal_rate", 1, True), (name + "-last_income", 1, True),
                   (name + "-last_marginal_rate", 1, True)]
        elif name == "SimpleLabor":
            a += [("SimpleLabor-skill", 1, True)]
    return a, p, pa


def flat_keys(env):
    a, p, pa = _component_keys(env)
    a.append(("time", 1, False))
    p.append(("time", 1, False))
    a_w, p_w, pa_w = env.world_flat_keys()
    a += a_w
    p += p_w
    pa += pa_w

    def table(items):
        out, off = [], 0
        for key, size, scalar in sorted(items):
            out.append((key, off, size, scalar))
            off += size
        return out, off

    ta, na = table(a)
    tp, npl = table(p)
    tpa, npa = table(pa)
    return {"a": ta, "p": tp, "pa": tpa, "sizes": {"a": na, "p": npl, "pa": npa}}


def mask_keys(env):
    """The flattened action masks as a key table: {"a": [(key, offset, size)], "p": [...], "sizes": {...}}.

    The reference's `_generate_masks` (base_env.py:706-756) collects one mask per action subspace -- key
    "<Component>" or "<Component>.<sub-action>" -- and `flatten_masks` (base_agent.py:440-460; the collated branch
    base_env.py:729-748) concatenates them in action-subspace order, with ONE leading NO-OP entry in single-action mode
    and one NO-OP entry in front of EVERY subspace in multi-action mode.  The kernels always write that flattened
    vector; `flatten_masks=False` hands out each subspace's slice (without the NO-OP entries) under its key.
    Pinned against the live reference in tests/test_obs_keys.py."""
    names_a, names_p = env.action_subspace_names()
    out, sizes = {}, {}
    for who, names, multi in (("a", names_a, env.multi_action_mode_agents), ("p", names_p, env.multi_action_mode_planner)):
        tab, off = [], 0
        if not multi and names:
            off = 1  # the single NO-OP entry in front
        for name, dim in names:
            if multi:
                off += 1  # this subspace's own NO-OP entry
            tab.append((name, off, int(dim)))
            off += int(dim)
        if not names:
            off = 1  # an agent class without actions: the lone NO-OP entry
        out[who] = tab
        sizes[who] = off
    out["sizes"] = sizes
    return out

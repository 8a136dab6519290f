#!/usr/bin/env python
"""Exports the DATA the COVID scenario is initialised from
(/root/reference/ai_economist/datasets/covid19_datasets/data_and_fitted_params/:
model_constants.json, fitted_params.json, real_world_data.npz -- measurements and fitted
coefficients, not code) into one npz inside the package, so that
`CovidAndEconomySimulation` can be constructed on machines without the reference.

    python oracle/export_covid_data.py
"""
import json
import os

import numpy as np

from ref_harness import REFERENCE_ROOT

SRC = os.path.join(REFERENCE_ROOT, "ai_economist/datasets/covid19_datasets/data_and_fitted_params")
DST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   "ai-economist_amd", "foundation", "scenarios", "covid19_data.npz")

mc = json.load(open(os.path.join(SRC, "model_constants.json")))
fp = json.load(open(os.path.join(SRC, "fitted_params.json")))
rw = np.load(os.path.join(SRC, "real_world_data.npz"))
out = {}
for k, v in mc.items():
    if isinstance(v, dict):  # US_STATE_IDX_TO_STATE_NAME: stored as a plain string array
        v = [v[str(i)] for i in range(len(v))]
    out["mc_" + k] = np.array(v)
for k, v in fp.items():
    if k == "settings":
        continue
    out["fp_" + k] = np.array(v)
for k in ["policy", "subsidy", "susceptible", "infected", "recovered", "vaccinated", "unemployed", "deaths"]:
    a = rw[k]
    out["rw_" + k] = a.astype(np.int8) if k == "policy" else a
np.savez_compressed(DST, **out)
print("wrote", DST, os.path.getsize(DST), "bytes", sorted(out.keys()))

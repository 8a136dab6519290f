#!/usr/bin/env python
"""Exports the reference's map layouts (DATA, not code:
/root/reference/ai_economist/foundation/scenarios/simple_wood_and_stone/map_txt/*.txt,
parsed exactly as layout_from_file.py:96-112 does) into one compact npz of uint8 grids
(0 empty, 1 Wood source, 2 Stone source, 3 Water) so that configs that name an
`env_layout_file` of the reference keep working on machines without the reference.

    python oracle/export_layouts.py
"""
import glob
import os

import numpy as np

from ref_harness import REFERENCE_ROOT

SRC = os.path.join(REFERENCE_ROOT, "ai_economist/foundation/scenarios/simple_wood_and_stone/map_txt")
DST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   "ai-economist_amd", "foundation", "scenarios", "layouts.npz")
CODE = {"W": 1, "S": 2, "@": 3}

out = {}
for path in sorted(glob.glob(os.path.join(SRC, "*.txt"))):
    rows = open(path).read().split(";")
    h = len(rows)
    w = max(len(r) for r in rows)
    g = np.zeros((h, w), np.uint8)
    for r, row in enumerate(rows):
        for c, sym in enumerate(row):
            g[r, c] = CODE.get(sym, 0)
    out[os.path.basename(path)] = g
    print(os.path.basename(path), g.shape, [(g == k).sum() for k in (1, 2, 3)])
np.savez_compressed(DST, **out)
print("wrote", DST, os.path.getsize(DST), "bytes")

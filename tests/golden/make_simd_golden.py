"""tests/golden/make_simd_golden.py -- known answers of the src/simd hook table, produced by the REFERENCE's own scalar
definitions (src/simd/distances_ref.cc compiled where it lies into oracle/_ref, driven by oracle/ref_simd.cpp) on the
seeded inputs of tests/simd_cases.py.  Run in the dev container (needs /root/reference):
    python tests/golden/make_simd_golden.py
writes tests/golden/simd/table.npz (outputs only; the inputs are regenerated from the seed)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import simd_cases as sc  # noqa: E402
from oracle import binding as ob  # noqa: E402

ref = ob.Ref()
blob = {}
for d in sc.DIMS:
    for name, v in sc.evaluate(ref, d).items():
        blob[f"d{d}/{name}"] = v
os.makedirs(os.path.join(HERE, "simd"), exist_ok=True)
np.savez_compressed(os.path.join(HERE, "simd", "table.npz"), **blob)
print(len(blob), "entries")

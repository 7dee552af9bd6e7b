"""Writes tests/golden/scipy_minima.json: the minimum scipy finds for every case of tests/scipy_pin_model.py and for the
ladybug49-sized problem (minutes: finite-difference Jacobians).  python tests/golden/make_scipy_minima.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))

import scipy_pin_model as M  # noqa: E402

out = {}
for name in M.CASES:
    prob, dof, loss, width, _ = M.case(name)
    out[name] = float(M.scipy_minimum(prob, dof, loss, width))
    print(name, repr(out[name]), flush=True)
out["ladybug49"] = float(M.scipy_minimum(M.ladybug(), 3, sparse=True))
print("ladybug49", repr(out["ladybug49"]), flush=True)
json.dump(out, open(os.path.join(HERE, "scipy_minima.json"), "w"), indent=1, sort_keys=True)

"""Generates tests/golden/fountain11_flat.npz from the reference fixture.

Run in the build container (needs /root/reference, which does not exist on the
GPU box):   python tests/golden/make_fountain11_golden.py

Source: /root/reference/data/sfm/fountain11.bin -- the only BA-relevant golden
data the reference ships: a real, already bundle-adjusted
Strecha fountain-11 reconstruction written by TheiaSfM itself (11 views, one
shared PINHOLE intrinsics group, 16 616 tracks, 75 022 observations).  The
archive is parsed with theiasfm_amd.io.read_theia_reconstruction and flattened
with the residual-set rules of BundleAdjustReconstruction
(bundle_adjustment.cc:66-80, bundle_adjuster.cc:102-180); no arithmetic is
applied, the arrays are the archive's own doubles.

Known answers recorded with it (SURVEY section 4 / BASELINE.md, reproduced by
the oracle): per-observation reprojection RMSE 0.442277 px and Ceres-style
cost 1/2 sum |r|^2 = 7337.480164.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from theiasfm_amd import io  # noqa: E402
from oracle import oracle  # noqa: E402

rec = io.read_theia_reconstruction("/root/reference/data/sfm/fountain11.bin")
prob = io.flatten_reconstruction(rec)
prob.save(os.path.join(HERE, "fountain11_flat.npz"))
cost, rmse, bad = oracle.cost(prob)
known = dict(num_cameras=prob.num_cameras, num_groups=prob.num_groups,
             num_points=prob.num_points, num_observations=prob.num_observations,
             cost=cost, rmse=rmse, invalid=bad,
             survey_cost=7337.480164, survey_rmse=0.442277)
json.dump(known, open(os.path.join(HERE, "fountain11_known.json"), "w"), indent=1)
print(known)

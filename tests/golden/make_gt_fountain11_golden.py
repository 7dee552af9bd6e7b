"""Generates tests/golden/gt_fountain11_positions.json from the reference's ground-truth fixture.

Run in the build container (needs /root/reference):   python tests/golden/make_gt_fountain11_golden.py

Source: /root/reference/data/sfm/gt_fountain11.bin -- the Strecha fountain-11 ground truth (metres) that the
reference's own estimator tests align their bundle-adjusted result to
(src/theia/sfm/incremental_reconstruction_estimator_test.cc:101-156: AlignReconstructions, then every camera
position within 1e-2 m).  Stored: the GT camera positions in the camera order of fountain11_flat.npz (matched by
view NAME, as FindCommonViewsByName does), and the names.  No arithmetic is applied.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from theiasfm_amd import io  # noqa: E402

gt = io.read_theia_reconstruction("/root/reference/data/sfm/gt_fountain11.bin")
rec = io.read_theia_reconstruction("/root/reference/data/sfm/fountain11.bin")
prob = io.flatten_reconstruction(rec)
id_to_name = {v: k for k, v in rec.view_names.items()}
names = [id_to_name[v] for v in prob.meta["view_ids"]]
assert sorted(names) == sorted(gt.view_names), "every view must be common (the reference asserts the same)"
pos = [[float(x) for x in gt.views[gt.view_names[n]].extrinsics[:3]] for n in names]
json.dump(dict(view_names=names, gt_positions_m=pos, tolerance_m=1e-2,
               source="data/sfm/gt_fountain11.bin; criterion incremental_reconstruction_estimator_test.cc:140-156"),
          open(os.path.join(HERE, "gt_fountain11_positions.json"), "w"), indent=1)
print(names)

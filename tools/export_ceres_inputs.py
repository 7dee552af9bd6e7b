#!/usr/bin/env python3
"""Writes the input archives of the Ceres golden recipe (tools/make_ceres_golden.md):
tests/golden/ceres/<name>_input.bin, cereal Reconstruction files any TheiaSfM build can load.

  python tools/export_ceres_inputs.py [tiny ladybug49 ...]          (default: tiny ladybug49)

The problems are the seeded synthetic configurations of theiasfm_amd/synth.py (bit-reproducible), so
the device / oracle side of the comparison regenerates them instead of reading these files back."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from theiasfm_amd import io, synth  # noqa: E402

out = os.path.join(ROOT, "tests", "golden", "ceres")
os.makedirs(out, exist_ok=True)
for name in (sys.argv[1:] or ["tiny", "ladybug49"]):
    prob = synth.config(name)
    path = os.path.join(out, f"{name}_input.bin")
    io.write_theia_reconstruction(path, io.reconstruction_from_problem(prob, image_size=(1000, 1000)))
    print(path, os.path.getsize(path), "bytes;", prob.num_cameras, "views", prob.num_points, "tracks",
          prob.num_observations, "observations")

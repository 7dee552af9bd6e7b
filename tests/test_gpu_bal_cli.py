"""-m gpu: a problem in the Bundle-Adjustment-in-the-Large text format end to end -- file -> read_bal (SURVEY App. D:
the conversion Theia's Bundler importer applies, read_bundler_files.cc:88-200) -> BA -> SetOutlierTracksToUnestimated
-> BA -> write_bal -- through tools/ba_bal.py, the command a user with a real BAL file runs.  No BAL file ships with
the reference and there is no network, so the file is written here from a synthetic scene in BAL's own conventions
(camera-frame -z forward, pixel = -f d(p) P_xy / P_z, per-view [f, k1, k2]); a few gross feature errors are planted so
that the filter has something to find.  The same problem, loaded the same way, goes through the oracle."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle
from theiasfm_amd import abi, io, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bal_file_through_the_command_line(tmp_path):
    prob = synth.make_problem(24, 3000, 14000, seed=19, scene="ring", spread=0.5)
    rng = np.random.default_rng(2)
    bad_tracks = rng.choice(prob.num_points, 12, replace=False)
    sel = np.isin(prob.obs_point, bad_tracks)
    prob.obs_xy[sel] += rng.normal(0.0, 40.0, (int(sel.sum()), 2))
    src, dst = str(tmp_path / "problem-24-3000-pre.txt"), str(tmp_path / "adjusted.txt")
    io.write_bal(src, prob)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ba_bal.py"), src, "--out", dst, "--iterations", "30",
                        "--no-inner-iterations", "--filter", "4.0", "1.0"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    out = p.stdout
    assert "read 24 cameras, 3000 points, 14000 observations" in out
    m1 = re.search(r"^BA: .*RMSE ([0-9.]+) -> ([0-9.]+) px", out, re.M)
    m2 = re.search(r"^BA without the flagged tracks: .*RMSE ([0-9.]+) -> ([0-9.]+) px", out, re.M)
    mf = re.search(r"filter: (\d+) tracks with bad reprojections", out)
    assert m1 and m2 and mf, out[-1500:]
    assert float(m1.group(2)) < float(m1.group(1))
    assert int(mf.group(1)) >= 10                      # the planted tracks (a planted track can fall under the bound by chance)
    assert float(m2.group(2)) < 0.7                    # without them: the pixel noise of the scene (0.5 px)
    # the first BA against the oracle on the problem as read_bal delivers it
    loaded = io.read_bal(src)
    st, s = oracle.solve(loaded, abi.default_options(point_dof=3, linear_solver_type=abi.DENSE_SCHUR, max_num_iterations=30,
                                                     use_inner_iterations=0))
    assert st == 0 and abs(s.final_rmse - float(m1.group(2))) <= 1e-4  # (the tool prints four decimals)
    # the adjusted file is a BAL file again
    back = io.read_bal(dst)
    assert back.num_cameras == 24 and back.num_observations < 14000

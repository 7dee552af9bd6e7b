"""GPU: SelectGoodTracksForBundleAdjustment on the Venice-sized heavy problem (for rocprofv3 --kernel-trace)."""
import sys
sys.path.insert(0, ".")
from theiasfm_amd import abi, lib, synth

P = synth.config("venice1778_heavy")
o = abi.default_options(point_dof=3, linear_solver_type=abi.ITERATIVE_SCHUR)
s = lib.Solver(P, o)
for i in range(3):
    sel, ln, err, ss = s.select_good_tracks(10, 100, 100)
    print("call ms", ss.seconds * 1e3, "selected", ss.num_selected)
s.close()

"""GPU diagnostic: batched track BA vs the oracle on a mid-size problem; prints where they differ."""
import sys
import numpy as np
sys.path.insert(0, ".")
from oracle import oracle
from theiasfm_amd import abi, lib, synth

P = synth.make_problem(200, 100000, 500000, seed=9, scene="ring", spread=0.12)
o = abi.default_options(point_dof=3, max_num_iterations=10)
R, D = P.copy(), P.copy()
to, io, c0o, c1o = oracle.adjust_tracks(R, o)
td, idv, c0d, c1d, ts = lib.adjust_tracks(D, o)
k = np.bincount(P.obs_point, minlength=P.num_points)
print("tracks", P.num_points, "k hist", np.bincount(k)[:12])
bad_t = np.flatnonzero(td != to)
bad_i = np.flatnonzero(idv != io)
print("termination mismatches", bad_t.size, "iteration mismatches", bad_i.size)
print("c0 max rel diff", np.max(np.abs(c0d - c0o) / np.maximum(c0o, 1e-300)))
rel = np.abs(c1d - c1o) / np.maximum(np.abs(c1o), 1e-12)
print("c1 rel diff: max", rel.max(), "count > 1e-9:", int((rel > 1e-9).sum()))
w = np.argsort(-rel)[:10]
for t in w:
    print(f" track {t} k={k[t]} term d/o {td[t]}/{to[t]} it {idv[t]}/{io[t]} c0 {c0o[t]:.6e} c1 d/o {c1d[t]:.12e}/{c1o[t]:.12e}")
for t in bad_i[:10]:
    print(f" itmis track {t} k={k[t]} term d/o {td[t]}/{to[t]} it {idv[t]}/{io[t]} c1 d/o {c1d[t]:.12e}/{c1o[t]:.12e}")
dp = np.abs(D.points - R.points).max(axis=1)
print("points max abs diff", dp.max(), "count > 1e-6:", int((dp > 1e-6).sum()))

"""-m gpu: the C ABI from several host threads at once (SURVEY 8(b) "Threading": distinct BundleAdjuster instances run
concurrently from pool threads on disjoint tracks, estimate_track.cc:166-204; the boundary must be re-entrant).  ctypes
releases the GIL for the duration of a call, so these are concurrent tmi_ba_solve / tmi_ba_adjust_tracks /
tmi_ba_solver_* calls on one device; every result must equal the serial one bit for bit."""
import threading

import numpy as np
import pytest

from theiasfm_amd import abi, lib, synth

pytestmark = pytest.mark.gpu

N = 8


def _problems():
    out = []
    for i in range(N):
        if i % 4 == 3:
            out.append(synth.make_problem(14, 700, 3600, seed=200 + i, scene="ring", spread=0.5, shared_group_size=2,
                                          intrinsics_to_optimize=abi.INTRINSICS_FOCAL_LENGTH | abi.INTRINSICS_RADIAL_DISTORTION))
        else:
            out.append(synth.make_problem(10 + 3 * i, 600 + 150 * i, 3000 + 900 * i, seed=100 + i, scene="ring", spread=0.5))
    return out


def _options(i):
    # exact (the co-resident dataflow Cholesky: big launches of one process are chained) and iterative solvers,
    # inner iterations, both operators -- side by side
    solver = (abi.DENSE_SCHUR, abi.ITERATIVE_SCHUR, abi.SPARSE_SCHUR, abi.ITERATIVE_SCHUR)[i % 4]
    return abi.default_options(linear_solver_type=solver, point_dof=3 + (i % 2), max_num_iterations=6,
                               use_inner_iterations=1 if i % 4 == 2 else 0,
                               schur_mode=abi.SCHUR_IMPLICIT if i % 8 == 5 else abi.SCHUR_AUTO)


def test_concurrent_one_shot_solves_equal_the_serial_ones():
    probs = _problems()
    serial = []
    for i, p in enumerate(probs):
        q = p.copy()
        st, s = lib.solve(q, _options(i))
        serial.append((st, s.final_cost, s.num_iterations, int(s.num_linear_solver_iterations), q))
    results = [None] * N
    errors = []

    def work(i):
        try:
            q = probs[i].copy()
            st, s = lib.solve(q, _options(i))
            results[i] = (st, s.final_cost, s.num_iterations, int(s.num_linear_solver_iterations), q)
        except Exception as exc:  # noqa: BLE001
            errors.append((i, repr(exc)))

    for _ in range(2):  # twice: the second round starts with warm module / allocator state
        threads = [threading.Thread(target=work, args=(i,)) for i in range(N)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        for i in range(N):
            a, b = serial[i], results[i]
            assert a[0] == b[0] == 0 and a[1] == b[1] and a[2] == b[2] and a[3] == b[3], (i, a[:4], b[:4])
            assert (a[4].extrinsics == b[4].extrinsics).all() and (a[4].points == b[4].points).all()
            assert (a[4].intrinsics == b[4].intrinsics).all()


def test_concurrent_track_adjustment_on_disjoint_tracks():
    """BundleAdjustTrack x N threads on ONE problem's disjoint track subsets (estimate_track.cc:238-246)"""
    prob = synth.make_problem(20, 4000, 20000, seed=77, scene="ring", spread=0.5)
    opts = abi.default_options(point_dof=4, max_num_iterations=10)
    whole = prob.copy()
    term_s, it_s, _, c1_s, _ = lib.adjust_tracks(whole, opts)
    subsets = [np.arange(i, prob.num_points, N) for i in range(N)]
    out = [None] * N
    errors = []

    def work(i):
        try:
            q = prob.copy()
            q.point_constant = np.ones(prob.num_points, dtype=np.uint8)  # everything but this thread's tracks is skipped
            q.point_constant[subsets[i]] = prob.point_constant[subsets[i]]
            term, it, _, c1, _ = lib.adjust_tracks(q, opts)
            out[i] = (term, it, c1, q.points)
        except Exception as exc:  # noqa: BLE001
            errors.append((i, repr(exc)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(N)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(N):
        term, it, c1, pts = out[i]
        sel = subsets[i]
        assert (term[sel] == term_s[sel]).all() and (it[sel] == it_s[sel]).all() and (c1[sel] == c1_s[sel]).all()
        assert (pts[sel] == whole.points[sel]).all()

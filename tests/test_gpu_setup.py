"""-m gpu: the structure built in HBM by sort / scan kernels (structure_gpu.h) against the host
builder (structure.cpp): every static array identical, hence bit-identical solves."""
import numpy as np
import pytest

from theiasfm_amd import abi, lib, synth

pytestmark = pytest.mark.gpu

NAMES = ["path", "slice_ptr", "pt_k", "pt_const", "obs_cam", "obs_xy", "obs_cpos", "cam_ptr", "ub_i", "ub_j",
         "urow_ptr", "ucol_ptr", "ucol_u", "spc_row", "spc_u0", "spc_rptr", "pair_ptr", "pair_i", "pair_j",
         "launch headers", "pt_orig", "n_order", "n_spc", "Nslots"]


def _problems():
    yield "ladybug49", synth.config("ladybug49"), {}
    # long tracks (wave-per-track kernels), constant tracks, an unobserved track, a constant camera,
    # a camera without a block (everything constant), observations in shuffled order
    p = synth.make_problem(40, 3000, 26000, seed=3, scene="ring", spread=0.5, heavy_tail=0.01)
    rng = np.random.default_rng(1)
    perm = rng.permutation(p.num_observations)
    p.obs_camera, p.obs_point, p.obs_xy = p.obs_camera[perm].copy(), p.obs_point[perm].copy(), p.obs_xy[perm].copy()
    p.point_constant[::11] = 1
    keep = p.obs_point != 17
    p.obs_camera, p.obs_point, p.obs_xy = p.obs_camera[keep].copy(), p.obs_point[keep].copy(), p.obs_xy[keep].copy()
    p.camera_flags[3] = abi.CAMERA_POSITION_CONSTANT
    p.camera_flags[5] = abi.CAMERA_POSITION_CONSTANT | abi.CAMERA_ORIENTATION_CONSTANT
    a, b = p.group_offset[5], p.group_offset[6]
    p.intrinsics_constant[a:b] = 1
    yield "edge cases", p, {}
    yield "implicit operator", synth.make_problem(30, 2000, 12000, seed=8, scene="ring", spread=0.4), dict(schur_mode=abi.SCHUR_IMPLICIT)
    yield "alamo", synth.config("alamo"), {}


@pytest.mark.parametrize("name,prob,extra", list(_problems()), ids=lambda x: x if isinstance(x, str) else "")
def test_device_built_structure_equals_host_built(name, prob, extra, monkeypatch):
    opts = abi.default_options(point_dof=3, linear_solver_type=abi.ITERATIVE_SCHUR, max_num_iterations=6,
                               use_inner_iterations=0, **extra)
    monkeypatch.setenv("TMI_BA_HOST_SETUP", "1")
    a = prob.copy()
    sh = lib.Solver(a, opts)
    cs_h = sh.structure_checksums()
    st_h, sum_h = sh.solve(opts)
    sh.download()
    sh.close()
    monkeypatch.delenv("TMI_BA_HOST_SETUP")
    b = prob.copy()
    sd = lib.Solver(b, opts)
    cs_d = sd.structure_checksums()
    st_d, sum_d = sd.solve(opts)
    sd.download()
    sd.close()
    assert cs_h[0] == 0 and cs_d[0] == 1
    for i in range(1, 24):
        assert cs_h[i] == cs_d[i], f"{name}: {NAMES[i]} differs between the host and the device builder"
    assert st_h == st_d == 0
    assert sum_h.final_cost == sum_d.final_cost and sum_h.num_iterations == sum_d.num_iterations
    np.testing.assert_array_equal(a.extrinsics, b.extrinsics)
    np.testing.assert_array_equal(a.points, b.points)


def test_device_setup_reports_bad_input():
    p = synth.make_problem(8, 100, 500, seed=2)
    q = p.copy()
    q.obs_point[7] = q.obs_point[6]
    q.obs_camera[7] = q.obs_camera[6]
    with pytest.raises(lib.EngineError):
        lib.Solver(q, abi.default_options(point_dof=3))
    r = p.copy()
    r.obs_camera[3] = 99
    with pytest.raises(lib.EngineError):
        lib.Solver(r, abi.default_options(point_dof=3))


@pytest.mark.parametrize("world", [2, 3])
def test_device_built_shards_equal_host_built(world, monkeypatch):
    """Sharded handles with the matrix-free operator (the default for world > 1) build their rank's
    layout on the device: the same slices, the same arrays as structure.cpp's deal."""
    p = synth.make_problem(40, 3000, 26000, seed=5, scene="ring", spread=0.5, heavy_tail=0.01)
    p.point_constant[::13] = 1
    keep = p.obs_point != 21  # an unobserved track: rank 0 answers for it
    p.obs_camera, p.obs_point, p.obs_xy = p.obs_camera[keep].copy(), p.obs_point[keep].copy(), p.obs_xy[keep].copy()
    opts = abi.default_options(point_dof=3, linear_solver_type=abi.ITERATIVE_SCHUR, schur_mode=abi.SCHUR_AUTO,
                               use_inner_iterations=0)
    total_tracks = 0
    for rank in range(world):
        monkeypatch.setenv("TMI_BA_HOST_SETUP", "1")
        sh = lib.Solver(p.copy(), opts, rank, world)
        cs_h = sh.structure_checksums()
        sh.close()
        monkeypatch.delenv("TMI_BA_HOST_SETUP")
        sd = lib.Solver(p.copy(), opts, rank, world)
        cs_d = sd.structure_checksums()
        sd.close()
        assert cs_h[0] == 0 and cs_d[0] == 1
        for i in range(1, 24):
            assert cs_h[i] == cs_d[i], f"rank {rank}/{world}: {NAMES[i]} differs between the host and the device builder"
    del total_tracks

"""CPU tests of the multi-GPU plan: tracks are sharded, every rank
sees the same reduced-matrix structure, the shards partition the observations
and the Schur pairs exactly -- so summing the per-rank reduced systems (the one
all-reduce per LM iteration) reproduces the single-rank system.  Plus the
torch.distributed hook itself with world_size 2 over gloo."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402
from theiasfm_amd import abi, lib, synth  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _built():
    entry.build_engine()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_shards_partition_tracks_observations_and_pairs(world):
    prob = synth.make_problem(40, 3000, 15000, seed=9, scene="ring", spread=0.3)
    prob.camera_flags[3] = abi.CAMERA_POSITION_CONSTANT | abi.CAMERA_ORIENTATION_CONSTANT
    prob.point_constant[::50] = 1
    a, b = prob.group_offset[3], prob.group_offset[4]
    prob.intrinsics_constant[a:b] = 1
    one = lib.structure_stats(prob, 0, 1)
    assert one["observations"] == prob.num_observations
    assert one["tracks"] == prob.num_points
    # camera 3: constant extrinsics AND constant intrinsics -> no reduced block at all
    assert one["reduced_blocks"] == prob.num_cameras - 1 and one["block_dim"] == 9
    parts = [lib.structure_stats(prob, r, world) for r in range(world)]
    for p in parts:  # the structure every rank all-reduces over is identical
        for k in ("reduced_blocks", "block_dim", "upper_blocks", "bsr_blocks", "block_checksum"):
            assert p[k] == one[k], k
    for k in ("tracks", "observations", "pairs", "observation_checksum", "pair_checksum"):
        assert sum(p[k] for p in parts) == one[k], k
    # balance: slices are dealt longest-work-first, work = pairs + 5 * observations
    work = np.array([p["pairs"] + 5.0 * p["observations"] for p in parts])
    assert work.max() / work.mean() < 1.15


@pytest.mark.parametrize("world", [2, 4, 8])
def test_matrix_free_dealing_balances_observations(world):
    """The default operator on several ranks is matrix-free, and its slices are dealt by observations (structure.cpp;
    ADVICE r5: the statistics call used to describe the pair-weighted dealing only).  A heavy-tailed problem is where the
    two rules differ: the rank that holds the longest tracks gets few observations under the pair rule."""
    prob = synth.make_problem(80, 8000, 44000, seed=21, scene="ring", spread=0.4, heavy_tail=0.02)
    one = lib.structure_stats(prob, 0, 1, forms_S=False)
    mf = [lib.structure_stats(prob, r, world, forms_S=False) for r in range(world)]
    ex = [lib.structure_stats(prob, r, world, forms_S=True) for r in range(world)]
    for parts in (mf, ex):  # both dealings partition the tracks and observations of the problem
        for k in ("tracks", "observations", "observation_checksum"):
            assert sum(p[k] for p in parts) == one[k], k
        for p in parts:
            assert p["reduced_blocks"] == one["reduced_blocks"] and p["block_dim"] == one["block_dim"]
    obs_mf = np.array([p["observations"] for p in mf], dtype=float)
    obs_ex = np.array([p["observations"] for p in ex], dtype=float)
    assert obs_mf.max() / obs_mf.mean() < 1.05 and obs_mf.min() / obs_mf.mean() > 0.95
    # ... which the pair-weighted dealing does not give on this problem (its balance is pairs + 5 x observations)
    assert obs_ex.min() / obs_ex.mean() < obs_mf.min() / obs_mf.mean()


def test_structure_rejects_bad_input():
    prob = synth.config("tiny")
    bad = prob.copy()
    bad.obs_camera[0] = 99
    with pytest.raises(lib.EngineError):
        lib.structure_stats(bad)
    dup = prob.copy()
    dup.obs_camera[1] = dup.obs_camera[0]  # same view twice in one track
    dup.obs_point[1] = dup.obs_point[0]
    with pytest.raises(lib.EngineError):
        lib.structure_stats(dup)
    with pytest.raises(lib.EngineError):
        lib.structure_stats(prob, rank=2, world=2)


def test_empty_and_ragged_problems():
    # tracks without observations and cameras without observations are tolerated
    prob = synth.make_problem(6, 50, 300, seed=2, scene="allsee")
    keep = prob.obs_point >= 10
    p2 = abi.Problem(prob.extrinsics, prob.camera_group, prob.camera_flags, prob.group_model,
                     prob.group_offset, prob.intrinsics, prob.intrinsics_constant, prob.points,
                     prob.point_constant, prob.obs_camera[keep], prob.obs_point[keep], prob.obs_xy[keep])
    s = lib.structure_stats(p2)
    assert s["tracks"] == 40 and s["observations"] == 240
    empty = abi.Problem(prob.extrinsics, prob.camera_group, prob.camera_flags, prob.group_model,
                        prob.group_offset, prob.intrinsics, prob.intrinsics_constant, prob.points,
                        prob.point_constant, prob.obs_camera[:0], prob.obs_point[:0], prob.obs_xy[:0])
    s = lib.structure_stats(empty)
    assert s["tracks"] == 0 and s["observations"] == 0 and s["upper_blocks"] == 0


def test_gloo_world2_allreduce_hook(tmp_path):
    """theiasfm_amd.dist.make_host_allreduce with two processes over gloo: the hook
    the engine calls must sum a raw buffer in place across ranks."""
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent(f"""
        import ctypes, os, sys
        sys.path.insert(0, {ROOT!r})
        import numpy as np
        from theiasfm_amd import dist
        rank, world, local = dist.init_from_env(backend="gloo")
        hook = dist.make_host_allreduce()
        buf = np.arange(1000, dtype=np.float64) * (rank + 1)
        assert hook(buf.ctypes.data, buf.size, 0) == 0
        assert np.array_equal(buf, np.arange(1000, dtype=np.float64) * 3.0), buf[:4]
        small = np.array([rank + 1.0])
        hook(small.ctypes.data, 1, 0)
        assert small[0] == 3.0
        print("rank", rank, "ok")
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs

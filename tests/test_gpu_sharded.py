"""-m gpu: the multi-GPU code path exercised on ONE device.

A gpurun box has a single GPU, so the sharded path is validated by emulation:
`world` solver instances (rank r of world, each owning its shard of the tracks)
run concurrently in threads on the same device; the all-reduce hook sums their
device buffers exactly where RCCL would.  The sharded solve must reproduce the
single-rank solve (same LM trajectory; sums differ only in association order).
A second test drives the real torch.distributed/RCCL hook with a 1-rank group."""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from theiasfm_amd import abi, dist, lib, synth

pytestmark = pytest.mark.gpu


class EmulatedAllReduce:
    def __init__(self, world):
        import torch
        self.torch = torch
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.calls = 0

    def hook(self, rank):
        torch = self.torch

        def fn(ptr, count, stream):
            torch.cuda.ExternalStream(stream).synchronize()
            self.slots[rank] = torch.as_tensor(dist._DevArray(ptr, count), device="cuda")
            self.barrier.wait()
            if rank == 0:
                total = self.slots[0].clone()
                for t in self.slots[1:]:
                    total += t
                for t in self.slots:
                    t.copy_(total)
                torch.cuda.synchronize()
                self.calls += 1
            self.barrier.wait()
            return 0
        return fn


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("solver_type,mode", [(abi.ITERATIVE_SCHUR, abi.SCHUR_EXPLICIT),
                                              (abi.ITERATIVE_SCHUR, abi.SCHUR_IMPLICIT),
                                              (abi.ITERATIVE_SCHUR, abi.SCHUR_AUTO),
                                              (abi.DENSE_SCHUR, abi.SCHUR_AUTO)])
@pytest.mark.parametrize("inner", [0, 1])
@pytest.mark.parametrize("shared", [False, True])
def test_sharded_solve_matches_single_rank(world, solver_type, mode, inner, shared):
    import torch
    torch.cuda.init()
    if shared:
        # free intrinsics shared by groups of three views, mixed camera models (BASELINE config 5)
        prob = synth.make_problem(18, 1200, 6000, seed=71, scene="ring", spread=0.4, shared_group_size=3,
                                  models=[(abi.PINHOLE, 0.5), (abi.PINHOLE_RADIAL_TANGENTIAL, 0.25),
                                          (abi.FISHEYE, 0.25)],
                                  intrinsics_to_optimize=abi.INTRINSICS_FOCAL_LENGTH | abi.INTRINSICS_RADIAL_DISTORTION)
    else:
        prob = synth.config("ladybug49")
    # inner = 1: the coordinate-descent sweep after each LM step; the per-view sums of its
    # camera-side blocks are all-reduced, its point set is local to the owning rank
    opts = abi.default_options(linear_solver_type=solver_type, point_dof=3, schur_mode=mode,
                               use_inner_iterations=inner)
    single = prob.copy()
    st, s1 = lib.solve(single, opts)
    assert st == 0

    emu = EmulatedAllReduce(world)
    shards = [prob.copy() for _ in range(world)]
    solvers = []
    for r in range(world):
        sv = lib.Solver(shards[r], opts, rank=r, world=world)
        sv.set_allreduce(emu.hook(r))
        solvers.append(sv)
    results = [None] * world

    def run(r):
        results[r] = solvers[r].solve(opts)

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert all(not t.is_alive() for t in threads)
    assert emu.calls > 0
    for r in range(world):
        st_r, s_r = results[r]
        assert st_r == 0 and s_r.success == 1
        # every rank reports the same global numbers
        assert s_r.num_iterations == s1.num_iterations
        assert s_r.num_successful_steps == s1.num_successful_steps
        assert abs(s_r.initial_cost - s1.initial_cost) <= 1e-12 * s1.initial_cost
        assert abs(s_r.final_cost - s1.final_cost) <= 1e-9 * s1.final_cost
        assert abs(s_r.final_rmse - s1.final_rmse) <= 1e-9
        assert s_r.final_cost == results[0][1].final_cost
    # cameras are replicated: identical on every rank and equal to the single-rank result
    merged = prob.copy()
    seen = np.zeros(prob.num_points, dtype=int)
    for r in range(world):
        out = solvers[r].download()
        assert np.abs(out.extrinsics - single.extrinsics).max() < 1e-6 * 100.0
        assert (out.extrinsics == solvers[0].problem.extrinsics).all()
        moved = np.any(out.points != prob.points, axis=1)
        seen += moved
        merged.points[moved] = out.points[moved]
        solvers[r].close()
    assert (seen == 1).all()            # every track is owned by exactly one rank
    assert np.abs(merged.points - single.points).max() < 1e-6 * 100.0


def _run_sharded(prob, opts, world):
    emu = EmulatedAllReduce(world)
    solvers = []
    for r in range(world):
        sv = lib.Solver(prob.copy(), opts, rank=r, world=world)
        sv.set_allreduce(emu.hook(r))
        solvers.append(sv)
    results = [None] * world

    def run(r):
        results[r] = solvers[r].solve(opts)

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert all(not t.is_alive() for t in threads)
    for sv in solvers:
        sv.close()
    return results


@pytest.mark.parametrize("mode", [abi.SCHUR_EXPLICIT, abi.SCHUR_IMPLICIT])
def test_sharded_ranks_leave_the_solve_together(mode):
    """Termination decisions on a sharded solve come from all-reduced data only: the gradient-tolerance
    vote rides in the scalar tail of the reduced system's all-reduce, the time limit in the tail of the
    trial-step scalars.  Every rank must stop in the same iteration for the same reason (a rank that left
    alone would leave the others in a collective)."""
    import torch
    torch.cuda.init()
    prob = synth.config("ladybug49")
    base = dict(linear_solver_type=abi.ITERATIVE_SCHUR, point_dof=3, schur_mode=mode, use_inner_iterations=0)
    # (a) a gradient tolerance every start point satisfies: zero iterations, as on one rank
    o = abi.default_options(gradient_tolerance=1e30, **base)
    st1, s1 = lib.solve(prob.copy(), o)
    assert st1 == 0 and s1.num_iterations == 0 and b"gradient tolerance" in s1.message
    for st_r, s_r in _run_sharded(prob, o, 2):
        assert st_r == 0 and s_r.num_iterations == 0 and b"gradient tolerance" in s_r.message
        assert s_r.final_cost == s1.final_cost or abs(s_r.final_cost - s1.final_cost) <= 1e-12 * s1.final_cost
    # (b) a tolerance reached after a few steps: same iteration count on every rank and on one rank
    o = None
    for tol in (1e2, 1e1, 1.0, 1e-1, 1e-2, 1e-3, 1e-4, 1e-5, 1e-6):
        o = abi.default_options(gradient_tolerance=tol, function_tolerance=-1.0, parameter_tolerance=-1.0,
                                max_num_iterations=40, **base)
        st1, s1 = lib.solve(prob.copy(), o)
        if st1 == 0 and b"gradient tolerance" in s1.message and 0 < s1.num_iterations < 40:
            break
    else:
        pytest.fail("no gradient tolerance in the ladder stops the single-rank solve after a few iterations")
    for st_r, s_r in _run_sharded(prob, o, 2):
        assert st_r == 0 and b"gradient tolerance" in s_r.message
        assert s_r.num_iterations == s1.num_iterations
    # (c) a time limit that has already passed when the first trial step is evaluated
    o = abi.default_options(max_solver_time_in_seconds=0.0, max_num_iterations=20, **base)
    res = _run_sharded(prob, o, 2)
    assert all(st_r == 0 for st_r, _ in res)
    assert len({int(s_r.num_iterations) for _, s_r in res}) == 1
    assert all(b"maximum solver time" in s_r.message for _, s_r in res)
    assert res[0][1].num_iterations <= 2


def test_rccl_hook_single_rank_group():
    """The production hook: torch.distributed backend nccl (= RCCL), a raw device
    pointer aliased as a tensor, enqueued on a foreign HIP stream."""
    import torch
    import torch.distributed as td
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if not td.is_initialized():
        torch.cuda.set_device(0)
        td.init_process_group(backend="nccl", rank=0, world_size=1)
    hook = dist.make_device_allreduce()
    buf = torch.arange(4096, dtype=torch.float64, device="cuda")
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        buf *= 2.0
    assert hook(buf.data_ptr(), buf.numel(), st.cuda_stream) == 0
    st.synchronize()
    assert torch.equal(buf.cpu(), torch.arange(4096, dtype=torch.float64) * 2.0)
    td.destroy_process_group()


def test_native_rccl_transport_single_rank():
    """The engine's own RCCL binding (dlopen'ed librccl, ncclCommInitRank, ncclAllReduce
    on the engine's stream) with a 1-rank communicator."""
    prob = synth.config("tiny")
    sv = lib.Solver(prob, abi.default_options(point_dof=3))
    uid = lib.rccl_unique_id()
    assert len(uid) == 128 and any(uid)
    sv.init_rccl(uid)
    assert sv.debug_allreduce(3.25) == 3.25
    # a hook is ignored once RCCL is bound
    sv.set_allreduce(lambda p, n, st: 1)
    assert sv.debug_allreduce(-1.5) == -1.5
    st, s = sv.solve(abi.default_options(point_dof=3, max_num_iterations=3))
    assert st == 0
    sv.close()


@pytest.mark.parametrize("mode,solver,inner,shared", [
    (abi.SCHUR_EXPLICIT, abi.ITERATIVE_SCHUR, 0, 0), (abi.SCHUR_AUTO, abi.ITERATIVE_SCHUR, 0, 0),
    (abi.SCHUR_EXPLICIT, abi.ITERATIVE_SCHUR, 1, 0), (abi.SCHUR_AUTO, abi.ITERATIVE_SCHUR, 1, 0),
    (abi.SCHUR_AUTO, abi.DENSE_SCHUR, 1, 0), (abi.SCHUR_AUTO, abi.ITERATIVE_SCHUR, 1, 1),
    (abi.SCHUR_EXPLICIT, abi.ITERATIVE_SCHUR, 0, 1)])
def test_two_processes_one_gpu(mode, solver, inner, shared):
    """Real separate processes (RANK / WORLD_SIZE from the environment, as torchrun sets
    them), each owning its shard of the tracks; they share cuda:0 so the sums travel
    through gloo (host-staged hook) instead of RCCL.  AUTO picks the implicit operator for world > 1.
    Both Schur operators, the exact solver, inner iterations (their per-view sums are all-reduced as well) and
    shared intrinsics blocks; every rank must end where the single-rank solve ends."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import mp_sharded_worker as worker
    prob = worker.problem(shared)
    single = prob.copy()
    st, s1 = lib.solve(single, worker.options(mode if solver != abi.ITERATIVE_SCHUR or mode == abi.SCHUR_EXPLICIT else abi.SCHUR_IMPLICIT,
                                              solver, inner))
    assert st == 0
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29540 + mode + 4 * inner + 8 * shared + (16 if solver != abi.ITERATIVE_SCHUR else 0)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "mp_sharded_worker.py"), str(mode), str(solver),
                               str(inner), str(shared)],
                              env=dict(env, RANK=str(r), LOCAL_RANK="0"), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    res = [json.loads([ln for ln in o.splitlines() if ln.startswith("RESULT ")][0][7:]) for o in outs]
    for r in res:
        assert r["status"] == 0 and r["iters"] == s1.num_iterations
        assert r["pcg"] == s1.num_linear_solver_iterations and r["inner_sweeps"] == s1.num_inner_iteration_steps
        assert abs(r["cost"] - s1.final_cost) <= 1e-9 * s1.final_cost
        assert abs(r["rmse"] - s1.final_rmse) <= 1e-9
        assert np.abs(np.array(r["ext"]) - single.extrinsics).max() < 1e-6 * 100.0
        assert np.abs(np.array(r["intr"]) - single.intrinsics).max() < 1e-6 * max(1.0, np.abs(single.intrinsics).max())
        if solver == abi.ITERATIVE_SCHUR:
            assert (r["pairs"] == 0) == (mode == abi.SCHUR_AUTO)
    assert res[0]["cost"] == res[1]["cost"] and res[0]["ext"] == res[1]["ext"]

"""Single-GPU probe of the per-rank cost of the REAL sharded solve at world sizes 1/2/4/8 (no second GPU needed).

Round 6: up to round 5 this probe ran ONE rank's shard with a hook that multiplied the all-reduce
buffer by `world` -- a sub-problem of its own, whose PCG counts differed from rank to rank.  Now, per world size:
  1. RECORD: all `world` ranks solve together on the one GPU (threads, a hook that really sums their buffers, as
     tests/test_gpu_sharded.py does) and every all-reduced buffer is kept on the device, in call order;
  2. REPLAY: the first and the last rank then solve ALONE, their hook copying the recorded sums into the buffer on the
     engine's stream where RCCL would deliver them -- the rank walks exactly the trajectory of the sharded solve (same
     LM / PCG iteration counts as the one-rank run) with the GPU to itself.
What is timed is per-rank compute + launches + host read-backs with a device-to-device copy standing in for every
collective: NO xGMI latency or bandwidth.  schur_mode auto (the matrix-free operator on several ranks); the explicit
mode's 0.83 GB all-reduce per LM iteration is not replayed (rounds 2-5: profiles/HISTORY.md).
usage: python tools/scale_probe.py [world ...]"""
import json
import os
import sys
import threading
import time

sys.path.insert(0, ".")
import torch  # noqa: E402

from theiasfm_amd import abi, dist, lib, synth  # noqa: E402

workload = os.environ.get("TMI_PROBE_WORKLOAD", "venice1778_heavy")
profile = int(os.environ.get("TMI_PROBE_PROFILE", "1"))  # 0: no per-class HIP events (the timing bench.py sees)
prob = synth.config(workload)
steps = 10
worlds = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
base = dict(point_dof=3, linear_solver_type=abi.ITERATIVE_SCHUR, function_tolerance=-1.0, gradient_tolerance=-1.0,
            parameter_tolerance=-1.0, schur_mode=0, use_inner_iterations=0)
o_warm = abi.default_options(max_num_iterations=2, **base)
o_time = abi.default_options(max_num_iterations=steps, profile_kernels=profile, **base)


class Recorder:
    """the summing hook of tests/test_gpu_sharded.py, keeping every sum"""

    def __init__(self, world):
        self.world, self.barrier, self.slots, self.log = world, threading.Barrier(world), [None] * world, []

    def hook(self, rank):
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
                self.log.append(total)
            self.barrier.wait()
            return 0
        return fn


def replay_hook(log):
    state = {"i": 0, "streams": {}, "tensors": {}}

    def fn(ptr, count, stream):
        t = state["tensors"].get((ptr, count))
        if t is None:
            t = state["tensors"][(ptr, count)] = torch.as_tensor(dist._DevArray(ptr, count), device="cuda")
        ext = state["streams"].get(stream)
        if ext is None:
            ext = state["streams"][stream] = torch.cuda.ExternalStream(stream)
        src = log[state["i"]]
        state["i"] += 1
        assert src.numel() == count, (src.numel(), count)
        with torch.cuda.stream(ext):
            t.copy_(src, non_blocking=True)
        return 0
    return fn, state


def report(world, rank_id, sm, el, extra):
    d = sm.as_dict()
    ks = {n: (l, round(1e3 * sec, 3)) for n, l, sec in zip(abi.KERNEL_CLASS_NAMES, d["kernel_launches"], d["kernel_seconds"]) if l}
    print(json.dumps(dict(workload=workload, profile_kernels=profile, world=world, rank=rank_id, schur_mode="auto",
                          its=int(sm.num_iterations), ms_per_iter=round(1e3 * el / max(1, sm.num_iterations), 3),
                          pcg=int(sm.num_linear_solver_iterations), final_cost=sm.final_cost,
                          kernel_ms_total=round(sum(v[1] for v in ks.values()), 2), kernels=ks, **extra)), flush=True)


one_rank = None
for world in worlds:
    if world == 1:
        s = lib.Solver(prob, o_warm, 0, 1)
        s.solve(o_warm)
        s.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st, sm = s.solve(o_time)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        one_rank = (int(sm.num_iterations), int(sm.num_linear_solver_iterations), sm.final_cost)
        report(1, 0, sm, el, dict(observations=int(prob.num_observations)))
        s.close()
        continue
    # ---- record: the sharded solve itself, all ranks on the one GPU
    solvers = [lib.Solver(prob, o_warm, r, world) for r in range(world)]
    logs = {}
    for name, opts in (("warm", o_warm), ("timed", o_time)):
        rec = Recorder(world)
        for r, sv in enumerate(solvers):
            sv.set_allreduce(rec.hook(r))
        if name == "timed":
            for sv in solvers:
                sv.reset()
        results = [None] * world

        def run(r, opts=opts, results=results):
            results[r] = solvers[r].solve(opts)
        threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=1200)
        assert all(not t.is_alive() for t in threads)
        logs[name] = (rec.log, results)
    sm_all = logs["timed"][1][0][1]
    sharded = (int(sm_all.num_iterations), int(sm_all.num_linear_solver_iterations), sm_all.final_cost)
    # ---- replay: first and last rank alone
    for rank_id in sorted({0, world - 1}):
        sv = solvers[rank_id]
        sv.reset()
        fn, state = replay_hook(logs["warm"][0])
        sv.set_allreduce(fn)
        sv.solve(o_warm)
        assert state["i"] == len(logs["warm"][0]), (state["i"], len(logs["warm"][0]))
        sv.reset()
        fn, state = replay_hook(logs["timed"][0])
        sv.set_allreduce(fn)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st, sm = sv.solve(o_time)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        assert state["i"] == len(logs["timed"][0])
        obs = int(lib.structure_stats(prob, rank_id, world, forms_S=False)["observations"])
        report(world, rank_id, sm, el, dict(observations=obs, real_sharded_problem=True,
                                           collectives_per_lm_iteration=round(len(logs["timed"][0]) / max(1, sharded[0]), 2),
                                           sharded_solve=dict(its=sharded[0], pcg=sharded[1], final_cost=sharded[2]),
                                           one_rank_solve=(dict(its=one_rank[0], pcg=one_rank[1], final_cost=one_rank[2]) if one_rank else None),
                                           replayed=dict(its=int(sm.num_iterations), pcg=int(sm.num_linear_solver_iterations),
                                                         final_cost=sm.final_cost)))
    for sv in solvers:
        sv.close()
    del logs
    torch.cuda.empty_cache()

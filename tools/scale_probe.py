"""Single-GPU probe of the per-rank cost at world sizes 1/2/4/8 (no second GPU needed):
rank 0 of `world` runs its shard, the all-reduce hook multiplies the buffer by `world` (stands in
for "the other ranks contribute about the same") on the engine's stream.  Numbers are per-rank
compute + launch + read-back time, WITHOUT real xGMI latency; results are not a solution."""
import json
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: E402

from theiasfm_amd import abi, dist, lib, synth  # noqa: E402

import os  # noqa: E402

workload = os.environ.get("TMI_PROBE_WORKLOAD", "venice1778_heavy")
profile = int(os.environ.get("TMI_PROBE_PROFILE", "1"))  # 0: no per-class HIP events (the timing bench.py sees)
prob = synth.config(workload)
steps = 10
worlds = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
# the shards differ in shape (rank 0 holds the longest tracks): every world size is timed on its first AND its last rank
cases = [(w, r, m) for w in worlds for r in sorted({0, w - 1}) for m in ((0,) if w == 1 else ((0, 1) if r == 0 else (0,)))]
for world, rank_id, mode in cases:
    if True:
        base = dict(point_dof=3, linear_solver_type=abi.ITERATIVE_SCHUR, function_tolerance=-1.0,
                    gradient_tolerance=-1.0, parameter_tolerance=-1.0, schur_mode=mode,
                    use_inner_iterations=0)
        o = abi.default_options(max_num_iterations=2, **base)
        s = lib.Solver(prob, o, rank_id, world)
        streams, tensors = {}, {}

        def hook(ptr, count, stream, world=world):
            t = tensors.get((ptr, count))
            if t is None:
                t = torch.as_tensor(dist._DevArray(ptr, count), device="cuda")
                tensors[(ptr, count)] = t
            ext = streams.get(stream)
            if ext is None:
                ext = streams[stream] = torch.cuda.ExternalStream(stream)
            with torch.cuda.stream(ext):
                t.mul_(float(world))
            return 0

        if world > 1:
            s.set_allreduce(hook)
        s.solve(o)
        s.reset()
        ot = abi.default_options(max_num_iterations=steps, profile_kernels=profile, **base)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st, sm = s.solve(ot)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        d = sm.as_dict()
        ks = {n: (l, round(1e3 * sec, 3)) for n, l, sec in zip(abi.KERNEL_CLASS_NAMES, d["kernel_launches"], d["kernel_seconds"]) if l}
        print(json.dumps(dict(workload=workload, profile_kernels=profile, world=world, rank=rank_id, observations=int(prob.num_observations) if world == 1 else int(lib.structure_stats(prob, rank_id, world)["observations"]) if mode == 1 else None, schur_mode=["auto", "explicit"][mode], its=int(sm.num_iterations),
                              ms_per_iter=round(1e3 * el / max(1, sm.num_iterations), 3),
                              pcg=int(sm.num_linear_solver_iterations), kernel_ms_total=round(sum(v[1] for v in ks.values()), 2),
                              kernels=ks)))
        s.close()

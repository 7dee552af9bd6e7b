"""-m gpu: bench.py's N > 1 branch, executed the way the driver launches it (bench.py's docstring):

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P \\
      bench.py --gpus 2 --steps K --warmup W

On the single-GPU test box both ranks have to share cuda:0, which RCCL refuses (one communicator rank per device),
so the sums travel through gloo and host memory (`--transport staged`: a functional check of the sharded path --
rendezvous, track sharding, one all-reduce hook call per exchange, barrier + max-over-ranks timing, rank 0's JSON
line -- not a measurement).  The RCCL transports are exercised as far as one device allows: a one-rank communicator
(tests/test_gpu_sharded.py) and, here, the native binding's clean failure when two ranks name the same device."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(n, port, *extra, timeout=900):
    cmd = [sys.executable]
    if n > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    cmd += [os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--no-extras", "--no-cpu-baseline"] + list(extra)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]  # rank 0 prints ONE JSON line
    return json.loads(lines[0])


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline")


@pytest.mark.parametrize("workload,steps", [("ladybug49", 4), ("venice1778_heavy", 2)])
def test_bench_two_ranks_staged_transport(workload, steps):
    args = ["--workload", workload, "--steps", str(steps), "--warmup", "1"]
    one = run_bench(1, 0, *args)
    two = run_bench(2, 29611 if workload == "ladybug49" else 29613, "--transport", "staged", *args)
    for k in CONTRACT_KEYS:
        assert k in two, k
    assert two["n_gpus"] == 2 and two["steps"] == steps and two["scaling"] == "strong" and two["metric"] == one["metric"]
    assert two["config"]["parallelism"] == "tracks sharded x2" and "gloo" in two["config"]["transport"]
    assert two["value"] > 0 and abs(two["value"] * two["ms_per_step"] * 1e-3 - two["config"]["observations"]) <= 1e-6 * two["config"]["observations"]
    # the same problem, the same iterations: the sharded solve ends where the single-rank one does
    assert two["config"]["observations"] == one["config"]["observations"]
    assert two["pcg_iterations"] == one["pcg_iterations"] and two["accepted_steps"] == one["accepted_steps"]
    assert abs(two["final_cost"] - one["final_cost"]) <= 1e-9 * one["final_cost"], (two["final_cost"], one["final_cost"])
    assert abs(two["final_rmse"] - one["final_rmse"]) <= 1e-9
    ar = two["allreduce"]
    assert ar["bytes_per_lm_iteration"] > 0 and ar["collectives_per_lm_iteration"] >= 2
    if workload == "venice1778_heavy":
        # ITERATIVE_SCHUR on several ranks: the matrix-free operator, the reduced vector all-reduced per PCG iteration
        assert two["config"]["schur_operator"].startswith("implicit") and two["schur_pairs"] == 0
        assert ar["bytes_per_lm_iteration"] < 16e6
    assert "allreduce" not in one


def test_bench_two_ranks_torch_hook_with_a_gloo_collective():
    """bench.py --transport torch_staged: dist.make_device_allreduce -- the hook `--transport torch` installs: device
    pointers aliased through __cuda_array_interface__, one cached tensor per engine buffer, the collective issued under
    the engine's ExternalStream -- driven end to end by two real processes; only the collective inside it is swapped for
    one that works with both ranks on cuda:0 (the first RCCL run must not also be the first
    execution of that code)."""
    args = ["--workload", "ladybug49", "--steps", "4", "--warmup", "1"]
    one = run_bench(1, 0, *args)
    two = run_bench(2, 29621, "--transport", "torch_staged", *args)
    assert two["n_gpus"] == 2 and "torch.distributed hook" in two["config"]["transport"]
    assert two["pcg_iterations"] == one["pcg_iterations"] and two["accepted_steps"] == one["accepted_steps"]
    assert abs(two["final_cost"] - one["final_cost"]) <= 1e-9 * one["final_cost"], (two["final_cost"], one["final_cost"])
    ar = two["allreduce"]
    assert len(ar["per_rank_ms_per_step"]) == 2 and all(x > 0 for x in ar["per_rank_ms_per_step"])
    assert abs(max(ar["per_rank_ms_per_step"]) - two["ms_per_step"]) <= 1e-3 * two["ms_per_step"]
    assert ar["torch_hook"]["calls"] > 0 and ar["torch_hook"]["bytes"] > 0
    assert ar["rank0_calls_per_lm_iteration_measured"] >= 2


def test_native_rccl_refuses_two_ranks_on_one_device_cleanly():
    """tmi_ba_solver_init_rccl with two ranks that both sit on cuda:0: ncclCommInitRank must come back with an error
    on both (no hang, no crash), the handle stays usable with the staged hook, and dist.init_native_rccl reports False
    on every rank so that a caller falls back."""
    worker = os.path.join(ROOT, "tests", "mp_rccl_same_device_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0",
               NCCL_DEBUG="WARN")
    procs = [subprocess.Popen([sys.executable, worker], env=dict(env, RANK=str(r), LOCAL_RANK="0"), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=180)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("RCCL initialisation with a duplicate device hung")
    assert all(p.returncode == 0 for p in procs), outs
    res = [json.loads([ln for ln in o.splitlines() if ln.startswith("RESULT ")][0][7:]) for o in outs]
    assert all(r["native"] is False and r["status"] == 0 for r in res), res
    assert res[0]["cost"] == res[1]["cost"]

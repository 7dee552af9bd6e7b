"""Where the set-up time of the drop-in call goes: theia::BundleAdjustReconstruction on the bench problem through
tools/e2e_bench with TMI_BA_SETUP_TIMING=1 (phases of the shim and of tmi_ba_solver_create on stderr).
usage: python tools/e2e_setup_probe.py [workload]"""
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, ".")
import __graft_entry__ as entry  # noqa: E402
import bench  # noqa: E402
from theiasfm_amd import synth  # noqa: E402

entry.build_host_shim()
prob = synth.config(sys.argv[1] if len(sys.argv) > 1 else "venice1778_heavy")
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "problem.bin")
    bench.write_problem_file(prob, path)
    p = subprocess.run([os.path.join("tools", "e2e_bench"), path, "10", "0", "2", "1"], capture_output=True, text=True,
                       env=dict(os.environ, TMI_BA_SETUP_TIMING="1"), timeout=900)
    print(p.stderr[-6000:])
    print(p.stdout[-1500:])

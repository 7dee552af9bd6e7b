"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol
include/theia_mi355_ba.h declares, struct layouts agree with the ctypes mirror,
defaults match BundleAdjustmentOptions (bundle_adjustment.h:78-122), and the
device path fails loudly (no CPU fallback) when no GPU is present."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402
from theiasfm_amd import abi, lib, synth  # noqa: E402


@pytest.fixture(scope="module")
def L():
    entry.build_engine()
    return lib.load()


def test_exports_every_declared_symbol(L):
    header = open(os.path.join(ROOT, "include", "theia_mi355_ba.h")).read()
    declared = set(re.findall(r"\b(tmi_ba_[a-z_]+)\s*\(", header))
    declared -= {"tmi_ba_allreduce_fn"}
    assert declared == set(lib.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name


def test_struct_layouts_match_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "theia_mi355_ba.h"\n'
        "int main(){printf(\"%zu %zu %zu %zu %zu %zu\\n\", sizeof(tmi_ba_problem),"
        "sizeof(tmi_ba_options), sizeof(tmi_ba_summary), offsetof(tmi_ba_options, point_dof),"
        "offsetof(tmi_ba_summary, kernel_seconds), offsetof(tmi_ba_problem, obs_xy));return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(abi.CProblem), C.sizeof(abi.COptions), C.sizeof(abi.CSummary),
            abi.COptions.point_dof.offset, abi.CSummary.kernel_seconds.offset,
            abi.CProblem.obs_xy.offset]
    assert got == want


def test_options_init_matches_reference_defaults(L):
    o = abi.COptions()
    L.tmi_ba_options_init(C.byref(o))
    d = abi.default_options()
    for name, _ in abi.COptions._fields_:
        if name == "iteration_trace":  # a pointer: NULL in both
            assert not getattr(o, name) and not getattr(d, name)
            continue
        assert getattr(o, name) == getattr(d, name), name
    # bundle_adjustment.h:78-122
    assert (o.loss_function_type, o.robust_loss_width) == (abi.LOSS_TRIVIAL, 2.0)
    assert (o.linear_solver_type, o.preconditioner_type) == (abi.SPARSE_SCHUR, abi.PRECOND_SCHUR_JACOBI)
    assert (o.num_threads, o.max_num_iterations, o.max_solver_time_in_seconds) == (1, 100, 3600.0)
    assert o.use_inner_iterations == 1
    assert (o.function_tolerance, o.gradient_tolerance, o.parameter_tolerance) == (1e-6, 1e-10, 1e-8)
    assert o.max_trust_region_radius == 1e12


def test_intrinsics_mask_matches_host_twin(L):
    for model in range(5):
        assert L.tmi_ba_intrinsics_size(model) == abi.INTRINSICS_SIZE[model]
        for bits in range(0x40):
            m = np.zeros(abi.INTRINSICS_SIZE[model], dtype=np.uint8)
            assert L.tmi_ba_intrinsics_constant_mask(model, bits, m.ctypes.data) == m.size
            assert (m == abi.intrinsics_constant_mask(model, bits)).all(), (model, bits)
    assert L.tmi_ba_intrinsics_size(7) == -1


def test_no_gpu_fails_loudly_no_cpu_fallback(L):
    if L.tmi_ba_device_count() > 0:
        pytest.skip("a GPU is visible")
    prob = synth.config("tiny")
    before = prob.copy()
    st, s = lib.solve(prob, abi.default_options())
    assert st == 2 and s.success == 0          # TMI_BA_ERR_NO_DEVICE
    assert (prob.points == before.points).all() and (prob.extrinsics == before.extrinsics).all()
    with pytest.raises(lib.EngineError):
        lib.Solver(prob, abi.default_options())


def test_null_arguments_are_rejected(L):
    s = abi.CSummary()
    assert L.tmi_ba_solve(None, None, C.byref(s)) == 1
    assert L.tmi_ba_solver_solve(None, None, None) == 1
    assert L.tmi_ba_status_string(5).decode().startswith("unsupported")

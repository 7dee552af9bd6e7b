"""GPU: the dense Cholesky kernels alone (tools/chol_harness.hip) against numpy: dataflow kernel and the launch-per-panel path.
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Itheiasfm_amd/csrc -o tools/libcholh.so tools/chol_harness.hip
  python tools/chol_harness_test.py [n ...]"""
import ctypes as C, numpy as np, sys, time
try:
    import torch  # noqa: F401  (one HIP runtime per process: torch ships its own libamdhip64)
except ImportError:
    pass
L = C.CDLL("tools/libcholh.so" if len(sys.argv) < 2 or not sys.argv[1].endswith(".so") else sys.argv[1])
def run(fn, A, b, reps=3):
    n = A.shape[0]; x = np.zeros(n); ms = C.c_double(); info = (C.c_int * 8)()
    rc = fn(A.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), n, reps, C.byref(ms), info)
    return rc, x, ms.value, list(info)
rng = np.random.default_rng(0)
for n in [int(a) for a in sys.argv[1:] if a.isdigit()] or [5, 63, 64, 65, 127, 128, 200, 441, 1000]:
    M = rng.normal(size=(n, max(n // 2, 4)))
    A = M @ M.T + np.diag(rng.uniform(0.5, 2.0, n)) * n * 0.05
    # block-sparse-like conditioning: scale rows/cols
    d = np.exp(rng.normal(0, 1.0, n)); A = A * d[:, None] * d[None, :]
    b = rng.normal(size=n)
    t = time.time(); xr = np.linalg.solve(A, b); tn = time.time() - t
    rc, x, ms, info = run(L.chol_df_solve, A, b)
    err = np.abs(x - xr).max() / np.abs(xr).max()
    res = np.linalg.norm(A @ x - b) / np.linalg.norm(b)
    rc2, x2, ms2, info2 = run(L.chol_panels_solve, A, b)
    err2 = np.abs(x2 - xr).max() / np.abs(xr).max()
    print(f"n={n:6d} df: {ms:8.3f} ms err {err:.2e} res {res:.2e} info {info[:4]} | panels: {ms2:8.3f} ms err {err2:.2e} | cond {np.linalg.cond(A):.1e} numpy {tn*1e3:.1f} ms", flush=True)

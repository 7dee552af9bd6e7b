"""CLUSTER_JACOBI (theia_mi355_ba.h, cluster_precond.h).  With shared intrinsics blocks a cluster is a shared block
together with the views that share it, inverted exactly; without them the views are clustered by visibility as Ceres'
VisibilityBasedPreconditioner does (CANONICAL_VIEWS / SINGLE_LINKAGE, bundle_adjustment.h:86-89) -- second half of
this file.

CPU: the oracle's restatement -- far fewer PCG iterations than SCHUR_JACOBI, the same LM trajectory within the
tolerance PCG's inexact solves allow, fall-back to SCHUR_JACOBI without shared blocks.
GPU: the device (gather from the blocks of S, batched tile-dataflow Cholesky, dataflow substitution per PCG iteration)
against the oracle: same iteration counts, cost 1e-9."""
import numpy as np
import pytest

from oracle import oracle
from theiasfm_amd import abi, synth

BITS = (abi.INTRINSICS_FOCAL_LENGTH | abi.INTRINSICS_PRINCIPAL_POINTS | abi.INTRINSICS_RADIAL_DISTORTION
        | abi.INTRINSICS_TANGENTIAL_DISTORTION)
MODELS = [(abi.PINHOLE, 0.5), (abi.PINHOLE_RADIAL_TANGENTIAL, 0.25), (abi.FISHEYE, 0.25)]


def shared_problem(n_views=60, groups=(2, 14), seed=5, per_view=800):
    # BASELINE config 5 in small: mixed models, shared groups of several sizes, every intrinsic but skew / aspect free
    return synth.make_problem(n_views, n_views * 160, n_views * per_view, seed=seed, scene="ring", spread=0.2, models=MODELS,
                              shared_group_sizes=groups, intrinsics_to_optimize=BITS)


def options(pre, **kw):
    o = dict(linear_solver_type=abi.ITERATIVE_SCHUR, point_dof=3, use_inner_iterations=0, preconditioner_type=pre,
             schur_mode=abi.SCHUR_EXPLICIT, max_num_iterations=8, function_tolerance=-1.0, gradient_tolerance=-1.0,
             parameter_tolerance=-1.0)
    o.update(kw)
    return abi.default_options(**o)


def test_oracle_cluster_jacobi_cuts_the_pcg_iterations():
    prob = shared_problem()
    res = {}
    for pre in (abi.PRECOND_SCHUR_JACOBI, abi.PRECOND_CLUSTER_JACOBI):
        p = prob.copy()
        st, s = oracle.solve(p, options(pre))
        assert st == 0 and s.success == 1
        res[pre] = (s, p)
    jac, clu = res[abi.PRECOND_SCHUR_JACOBI][0], res[abi.PRECOND_CLUSTER_JACOBI][0]
    assert clu.num_linear_solver_iterations * 3 < jac.num_linear_solver_iterations
    # both solve the same reduced systems to the same forcing tolerance: the trajectories agree to what that leaves
    assert abs(clu.final_cost - jac.final_cost) < 1e-4 * jac.final_cost
    assert clu.num_successful_steps == jac.num_successful_steps
    # CLUSTER_TRIDIAGONAL (round 6): the clusters plus the blocks between neighbours of the spanning forest -- at least
    # as good a preconditioner on this problem, the same minimum
    st, tri = oracle.solve(prob.copy(), options(abi.PRECOND_CLUSTER_TRIDIAGONAL))
    assert st == 0 and tri.success == 1
    assert tri.num_linear_solver_iterations <= clu.num_linear_solver_iterations + 2
    assert abs(tri.final_cost - jac.final_cost) < 1e-4 * jac.final_cost and tri.num_successful_steps == jac.num_successful_steps


def test_single_linkage_without_similar_views_is_the_merged_block_jacobi():
    """SINGLE_LINKAGE joins views whose visibility similarity is >= 0.9; the ring scene has none, every cluster is one
    view and its exact inverse is the view's whole block: SCHUR_JACOBI with merged blocks, to the bit"""
    prob = synth.config("ladybug49")
    outs = []
    for pre, kw in ((abi.PRECOND_SCHUR_JACOBI, {}), (abi.PRECOND_CLUSTER_JACOBI, dict(visibility_clustering_type=abi.SINGLE_LINKAGE))):
        p = prob.copy()
        st, s = oracle.solve(p, options(pre, max_num_iterations=5, **kw))
        assert st == 0
        outs.append((s.final_cost, s.num_linear_solver_iterations, p.extrinsics.copy()))
    assert abs(outs[0][0] - outs[1][0]) <= 1e-12 * outs[0][0] and outs[0][1] == outs[1][1]
    assert np.abs(outs[0][2] - outs[1][2]).max() <= 1e-10
    assert len(set(oracle.last_visibility_clusters(prob.num_cameras).tolist())) == prob.num_cameras


# ---- clusters by visibility (no shared intrinsics blocks) ------------------------------------------------------------
def schur_complement_graph(prob):
    """Ceres' CreateSchurComplementGraph on the PARAMETER blocks, literally: vertex 2 c = extrinsics of camera c, 2 c + 1 =
    its (private, free) intrinsics; weight |points both see| / sqrt(|points of a| |points of b|), self edges 1"""
    nc = prob.num_cameras
    free_intr = [bool((prob.intrinsics_constant[prob.group_offset[g]:prob.group_offset[g + 1]] == 0).any())
                 for g in prob.camera_group]
    verts = [2 * c for c in range(nc)] + [2 * c + 1 for c in range(nc) if free_intr[c]]
    vis = {v: set() for v in verts}
    for c, p in zip(prob.obs_camera, prob.obs_point):
        if prob.point_constant[p]:
            continue
        vis[2 * c].add(int(p))
        if free_intr[c]:
            vis[2 * c + 1].add(int(p))
    by_point = {}
    for v in verts:
        for p in vis[v]:
            by_point.setdefault(p, []).append(v)
    count = {}
    for vs in by_point.values():
        for a in vs:
            for b in vs:
                if a < b:
                    count[(a, b)] = count.get((a, b), 0) + 1
    w = {(v, v): 1.0 for v in verts}
    nbr = {v: {v} for v in verts}
    for (a, b), n in count.items():
        w[(a, b)] = w[(b, a)] = n / np.sqrt(len(vis[a]) * len(vis[b]))
        nbr[a].add(b)
        nbr[b].add(a)
    return sorted(verts), nbr, w


def canonical_views(verts, nbr, w, size_penalty=3.0, similarity_penalty=0.0, min_views=3):
    """canonical_views_clustering.cc, literally, with the preconditioner's constants; ties to the lower vertex"""
    sim, assigned, centers, valid = {}, {}, [], list(verts)
    while valid:
        best, best_d = None, -np.inf
        for v in valid:
            d = sum(max(0.0, w[(n, v)] - sim.get(n, 0.0)) for n in nbr[v]) - size_penalty
            d -= similarity_penalty * sum(w.get((c, v), 0.0) for c in centers)
            if d > best_d:
                best, best_d = v, d
        if best_d <= 0 and len(centers) >= min_views:
            break
        centers.append(best)
        valid.remove(best)
        for n in sorted(nbr[best]):
            if w[(n, best)] > sim.get(n, 0.0):
                sim[n], assigned[n] = w[(n, best)], best
    cid = {c: i for i, c in enumerate(centers)}
    return {v: (cid[assigned[v]] if v in assigned else v % len(centers)) for v in verts}


def same_partition(a, b):
    """two labelings of the same items describe the same partition"""
    m = {}
    for x, y in zip(a, b):
        if m.setdefault(x, y) != y:
            return False
    return len(set(m.values())) == len(m)


@pytest.mark.parametrize("name", ["ladybug49", "constant_blocks"])
def test_oracle_canonical_views_clusters_are_the_literal_algorithms(name):
    """the oracle keeps its clustering state per VIEW (the blocks of a view always share a cluster); the literal
    algorithm on the parameter-block graph, written here in Python from Ceres 1.14's sources as published, must give
    the same partition of the cameras"""
    prob = synth.config("ladybug49") if name == "ladybug49" else synth.make_problem(20, 1500, 7000, seed=23, scene="ring", spread=0.6)
    if name == "constant_blocks":
        prob.set_intrinsics_to_optimize(abi.INTRINSICS_NONE)              # extrinsics vertices only
        prob.camera_flags[3] = abi.CAMERA_POSITION_CONSTANT | abi.CAMERA_ORIENTATION_CONSTANT   # a view without a block
        prob.point_constant[::4] = 1                                       # constant points are no e-blocks
    st, _ = oracle.solve(prob.copy(), options(abi.PRECOND_CLUSTER_JACOBI, max_num_iterations=1))
    assert st == 0
    got = oracle.last_visibility_clusters(prob.num_cameras)
    verts, nbr, w = schur_complement_graph(prob)
    if name == "constant_blocks":
        verts = [v for v in verts if v != 6]
        nbr = {v: {n for n in ns if n != 6} for v, ns in nbr.items() if v != 6}
    want = canonical_views(verts, nbr, w)
    cams = [c for c in range(prob.num_cameras) if 2 * c in want]
    assert all(got[c] >= 0 for c in cams) and (name != "constant_blocks" or got[3] == -1)
    # extrinsics and intrinsics of a view: one cluster
    assert all(want[2 * c] == want[2 * c + 1] for c in cams if 2 * c + 1 in want)
    assert same_partition([int(got[c]) for c in cams], [want[2 * c] for c in cams])
    assert 3 <= len(set(want.values())) < len(cams)


def test_oracle_single_linkage_joins_views_that_see_the_same_tracks():
    """two views with (almost) identical visibility -- similarity >= 0.9 -- land in one cluster, nobody else does"""
    prob = synth.make_problem(12, 900, 4200, seed=31, scene="ring", spread=0.6)
    # view 7 becomes a second look at the tracks of view 2: drop 7's observations, copy 2's (a different pose sees them too)
    keep = prob.obs_camera != 7
    dup = prob.obs_camera == 2
    prob.obs_camera = np.concatenate([prob.obs_camera[keep], np.full(int(dup.sum()), 7, dtype=prob.obs_camera.dtype)])
    prob.obs_point = np.concatenate([prob.obs_point[keep], prob.obs_point[dup]])
    prob.obs_xy = np.concatenate([prob.obs_xy[keep], prob.obs_xy[dup]])
    st, s = oracle.solve(prob.copy(), options(abi.PRECOND_CLUSTER_JACOBI, max_num_iterations=2,
                                              visibility_clustering_type=abi.SINGLE_LINKAGE))
    assert st == 0
    cl = oracle.last_visibility_clusters(prob.num_cameras)
    assert cl[2] == cl[7] and len(set(cl.tolist())) == prob.num_cameras - 1


def test_oracle_visibility_clusters_precondition_better_than_block_jacobi():
    prob = synth.make_problem(60, 9000, 50000, seed=17, scene="ring", spread=0.3)
    res = {}
    for pre in (abi.PRECOND_SCHUR_JACOBI, abi.PRECOND_CLUSTER_JACOBI):
        st, s = oracle.solve(prob.copy(), options(pre, max_num_iterations=6))
        assert st == 0 and s.success == 1
        res[pre] = s
    jac, clu = res[abi.PRECOND_SCHUR_JACOBI], res[abi.PRECOND_CLUSTER_JACOBI]
    assert clu.num_linear_solver_iterations < jac.num_linear_solver_iterations
    assert abs(clu.final_cost - jac.final_cost) < 1e-4 * jac.final_cost and clu.num_successful_steps == jac.num_successful_steps


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["ladybug49_canonical", "ladybug49_auto_mode", "bigger_canonical_huber_dof4", "single_linkage_pair"])
def test_device_visibility_clusters_match_oracle(case):
    from theiasfm_amd import lib
    kw = dict(function_tolerance=1e-6, gradient_tolerance=1e-10, parameter_tolerance=1e-8, max_num_iterations=12)
    if case.startswith("ladybug49"):
        prob = synth.config("ladybug49")
        if case == "ladybug49_auto_mode":
            kw.update(schur_mode=abi.SCHUR_AUTO)  # auto must pick the formed S: the clusters are its submatrices
    elif case == "bigger_canonical_huber_dof4":
        prob = synth.make_problem(150, 24000, 130000, seed=29, scene="ring", spread=0.3)
        kw.update(loss_function_type=abi.LOSS_HUBER, robust_loss_width=3.0, point_dof=4)
    else:
        prob = synth.make_problem(12, 900, 4200, seed=31, scene="ring", spread=0.6)
        keep, dup = prob.obs_camera != 7, prob.obs_camera == 2
        prob.obs_camera = np.concatenate([prob.obs_camera[keep], np.full(int(dup.sum()), 7, dtype=prob.obs_camera.dtype)])
        prob.obs_point = np.concatenate([prob.obs_point[keep], prob.obs_point[dup]])
        prob.obs_xy = np.concatenate([prob.obs_xy[keep], prob.obs_xy[dup]])
        kw.update(visibility_clustering_type=abi.SINGLE_LINKAGE)
    o = options(abi.PRECOND_CLUSTER_JACOBI, **kw)
    a, b = prob.copy(), prob.copy()
    st_d, s_d = lib.solve(a, o)
    st_o, s_o = oracle.solve(b, o)
    assert st_d == st_o == 0, (s_d.message, s_o.message)
    assert s_d.num_matrix_free_iterations == 0
    assert s_d.num_iterations == s_o.num_iterations and s_d.num_successful_steps == s_o.num_successful_steps
    assert abs(int(s_d.num_linear_solver_iterations) - int(s_o.num_linear_solver_iterations)) <= 1
    assert abs(s_d.final_cost - s_o.final_cost) <= 1e-9 * s_o.final_cost
    assert np.abs(a.extrinsics - b.extrinsics).max() <= 1e-6 * 100.0
    # and it is not block-Jacobi in disguise
    st_j, s_j = lib.solve(prob.copy(), options(abi.PRECOND_SCHUR_JACOBI, **{k: v for k, v in kw.items() if k != "visibility_clustering_type"}))
    if case != "single_linkage_pair":
        assert s_d.num_linear_solver_iterations < s_j.num_linear_solver_iterations


@pytest.mark.gpu
def test_device_visibility_clusters_with_the_matrix_free_operator_fall_back_to_block_jacobi():
    """the clusters are submatrices of the formed S: a handle created for the matrix-free operator keeps the SCHUR_JACOBI
    blocks (as on several ranks) instead of failing the create"""
    from theiasfm_amd import lib
    prob = synth.config("ladybug49")
    kw = dict(function_tolerance=1e-6, gradient_tolerance=1e-10, parameter_tolerance=1e-8, max_num_iterations=12,
              schur_mode=abi.SCHUR_IMPLICIT)
    st_c, s_c = lib.solve(prob.copy(), options(abi.PRECOND_CLUSTER_JACOBI, **kw))
    st_j, s_j = lib.solve(prob.copy(), options(abi.PRECOND_SCHUR_JACOBI, **kw))
    assert st_c == st_j == 0
    assert s_c.num_linear_solver_iterations == s_j.num_linear_solver_iterations and s_c.final_cost == s_j.final_cost
    # ... and the caller is told which preconditioner it got: the summary names the one that ran
    assert s_c.effective_preconditioner_type == abi.PRECOND_SCHUR_JACOBI == s_j.effective_preconditioner_type
    st_e, s_e = lib.solve(prob.copy(), options(abi.PRECOND_CLUSTER_JACOBI, max_num_iterations=3))  # schur_mode explicit
    assert st_e == 0 and s_e.effective_preconditioner_type == abi.PRECOND_CLUSTER_JACOBI
    st_p, s_p = lib.solve(prob.copy(), options(abi.PRECOND_SCHUR_JACOBI_PARAMETER_BLOCKS, max_num_iterations=3))
    assert st_p == 0 and s_p.effective_preconditioner_type == abi.PRECOND_SCHUR_JACOBI_PARAMETER_BLOCKS
    st_x, s_x = lib.solve(prob.copy(), abi.default_options(linear_solver_type=abi.DENSE_SCHUR, point_dof=3, use_inner_iterations=0,
                                                           max_num_iterations=3))
    assert st_x == 0 and s_x.effective_preconditioner_type == 0  # exact solver: no preconditioner ran


@pytest.mark.gpu
def test_device_rejects_a_changed_clustering_type_and_the_other_cluster_preconditioner():
    """the visibility clusters -- and for CLUSTER_TRIDIAGONAL the chains of clusters -- are built at create: a solve with
    another visibility_clustering_type, or with the other one of the two cluster preconditioners, is rejected"""
    from theiasfm_amd import lib
    prob = synth.config("ladybug49")
    h = lib.Solver(prob, options(abi.PRECOND_CLUSTER_JACOBI))
    try:
        st, s = h.solve(options(abi.PRECOND_CLUSTER_JACOBI, visibility_clustering_type=abi.SINGLE_LINKAGE))
        assert st != 0 and b"visibility_clustering_type" in bytes(s.message)
        st, s = h.solve(options(abi.PRECOND_CLUSTER_TRIDIAGONAL))
        assert st == abi.ERR_INVALID_ARGUMENT and b"created for the other one" in bytes(s.message)
    finally:
        h.close()
    h = lib.Solver(prob, options(abi.PRECOND_CLUSTER_TRIDIAGONAL))
    try:
        st, s = h.solve(options(abi.PRECOND_CLUSTER_JACOBI))
        assert st == abi.ERR_INVALID_ARGUMENT
        st, s = h.solve(options(abi.PRECOND_CLUSTER_TRIDIAGONAL, max_num_iterations=2))
        assert st == 0 and s.effective_preconditioner_type == abi.PRECOND_CLUSTER_TRIDIAGONAL
    finally:
        h.close()


# ---- CLUSTER_TRIDIAGONAL (cluster_chains.h; oracle: tridiagonal_segments) -----------------------------------------------
def literal_tridiagonal_segments(prob, cluster_of_cam):
    """Ceres 1.14's ComputeClusterTridiagonalSparsity on the clusters handed in, written from its sources as published
    (visibility_based_preconditioner.cc: CreateClusterGraph -- edge weight = number of e-blocks both clusters see --,
    graph_algorithms.h: Degree2MaximumSpanningForest -- pairs <weight, <v1, v2>> sorted with reverse iterators, an edge
    is skipped when an end has degree 2 or the ends are connected) plus this repository's conventions: clusters numbered
    by their lowest camera, paths walked from their lower-numbered end.  Returns {camera: (segment, position)}."""
    cams = [c for c in range(prob.num_cameras) if cluster_of_cam[c] >= -1 and prob.camera_flags[c] != (abi.CAMERA_POSITION_CONSTANT | abi.CAMERA_ORIENTATION_CONSTANT)]
    raw = {}
    nxt = max([cluster_of_cam[c] for c in cams] + [-1]) + 1
    for c in cams:
        if cluster_of_cam[c] >= 0:
            raw[c] = int(cluster_of_cam[c])
        else:
            raw[c] = nxt
            nxt += 1
    renum = {}
    for c in cams:
        renum.setdefault(raw[c], len(renum))
    cl = {c: renum[raw[c]] for c in cams}
    ncl = len(renum)
    seen = [set() for _ in range(ncl)]
    for cam, pt in zip(prob.obs_camera.tolist(), prob.obs_point.tolist()):
        if cam in cl and not prob.point_constant[pt]:
            seen[cl[cam]].add(pt)
    edges = []
    for a in range(ncl):
        for b in range(a + 1, ncl):
            w = len(seen[a] & seen[b])
            if w > 0:
                edges.append((float(w), (a, b)))
    edges.sort(reverse=True)
    deg = [0] * ncl
    nb = [[] for _ in range(ncl)]
    comp = list(range(ncl))

    def find(a):
        while comp[a] != a:
            a = comp[a]
        return a
    for _, (a, b) in edges:
        if deg[a] == 2 or deg[b] == 2 or find(a) == find(b):
            continue
        nb[a].append(b)
        nb[b].append(a)
        deg[a] += 1
        deg[b] += 1
        ra, rb = sorted((find(a), find(b)))
        comp[rb] = ra
    out, visited, seg = {}, [False] * ncl, 0
    for c0 in range(ncl):
        if visited[c0] or deg[c0] == 2:
            continue
        path, prev, cur = [], -1, c0
        while cur >= 0:
            visited[cur] = True
            path.append(cur)
            nx = [x for x in nb[cur] if x != prev]
            prev, cur = cur, (nx[0] if nx else -1)
        members = [(c, q) for q, k in enumerate(path) for c in cams if cl[c] == k]
        if len(members) >= 2:
            for c, q in members:
                out[c] = (seg, q)
            seg += 1
    return out


@pytest.mark.parametrize("name", ["ladybug49", "ring150", "constant_blocks"])
def test_oracle_tridiagonal_segments_are_the_literal_algorithm(name):
    if name == "ladybug49":
        prob = synth.config("ladybug49")
    elif name == "ring150":
        prob = synth.make_problem(150, 6000, 30000, seed=29, scene="ring", spread=0.15)
    else:
        prob = synth.make_problem(20, 1500, 7000, seed=23, scene="ring", spread=0.6)
        prob.camera_flags[3] = abi.CAMERA_POSITION_CONSTANT | abi.CAMERA_ORIENTATION_CONSTANT
        prob.set_intrinsics_to_optimize(abi.INTRINSICS_NONE)
        prob.point_constant[::4] = 1
    st, s = oracle.solve(prob.copy(), options(abi.PRECOND_CLUSTER_TRIDIAGONAL, max_num_iterations=1))
    assert st == 0 and s.success == 1
    clusters = oracle.last_visibility_clusters(prob.num_cameras)
    seg, pos = oracle.last_tridiagonal_segments(prob.num_cameras)
    want = literal_tridiagonal_segments(prob, clusters)
    got = {c: (int(seg[c]), int(pos[c])) for c in range(prob.num_cameras) if seg[c] >= 0}
    assert got == want
    n_clusters = len(set(int(x) for x in clusters if x >= 0))
    assert n_clusters >= 3 and max(p for _, p in want.values()) >= 1  # a chain of several clusters exists


def test_oracle_tridiagonal_of_two_clusters_is_the_exact_inverse():
    """two shared intrinsics groups that hold every view: the chain of the two clusters with the blocks between them is S
    itself, so the preconditioned system is the identity and conjugate gradients stops at its second iteration"""
    prob = shared_problem(n_views=20, groups=(10, 10), seed=11, per_view=400)
    assert len(set(prob.camera_group.tolist())) == 2
    st, s = oracle.solve(prob.copy(), options(abi.PRECOND_CLUSTER_TRIDIAGONAL, max_num_iterations=5))
    assert st == 0 and s.success == 1
    assert s.num_linear_solver_iterations <= 2 * s.num_iterations
    st, j = oracle.solve(prob.copy(), options(abi.PRECOND_CLUSTER_JACOBI, max_num_iterations=5))
    assert st == 0 and j.num_linear_solver_iterations > s.num_linear_solver_iterations
    assert abs(s.final_cost - j.final_cost) < 1e-4 * j.final_cost


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["ladybug49", "ring150_huber_dof4", "shared_mixed_sizes", "shared_two_groups", "auto_mode"])
def test_device_cluster_tridiagonal_matches_oracle(case):
    from theiasfm_amd import lib
    kw = dict(function_tolerance=1e-6, gradient_tolerance=1e-10, parameter_tolerance=1e-8, max_num_iterations=10)
    if case == "ladybug49":
        prob = synth.config("ladybug49")
    elif case == "ring150_huber_dof4":
        prob = synth.make_problem(150, 24000, 130000, seed=29, scene="ring", spread=0.3)
        kw.update(loss_function_type=abi.LOSS_HUBER, robust_loss_width=3.0, point_dof=4)
    elif case == "shared_mixed_sizes":
        prob = shared_problem()
    elif case == "shared_two_groups":
        prob = shared_problem(n_views=20, groups=(10, 10), seed=11, per_view=400)
    else:
        prob = shared_problem(n_views=40, groups=(2, 9), seed=3)
        kw.update(schur_mode=abi.SCHUR_AUTO)  # auto must form S: the chains need the blocks between clusters
    a, b = prob.copy(), prob.copy()
    st_d, s_d = lib.solve(a, options(abi.PRECOND_CLUSTER_TRIDIAGONAL, **kw))
    st_o, s_o = oracle.solve(b, options(abi.PRECOND_CLUSTER_TRIDIAGONAL, **kw))
    assert st_d == st_o == 0 and s_d.success == s_o.success == 1
    assert s_d.effective_preconditioner_type == abi.PRECOND_CLUSTER_TRIDIAGONAL
    assert s_d.num_iterations == s_o.num_iterations and s_d.num_successful_steps == s_o.num_successful_steps
    assert s_d.num_linear_solver_iterations == s_o.num_linear_solver_iterations
    assert abs(s_d.final_cost - s_o.final_cost) <= 1e-9 * s_o.final_cost
    if case == "shared_two_groups":
        assert s_d.num_linear_solver_iterations <= 2 * s_d.num_iterations


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["retry", "failure"])
def test_device_cluster_tridiagonal_retry_and_failure_follow_the_oracle(path, monkeypatch):
    """Dropping the blocks between distant clusters can cost positive definiteness; Ceres then halves the off-diagonal
    cluster-pair cells and factors again, and a second failure fails the linear solve (the LM step is invalid).  The two
    attempts' factors are test hooks on both sides: 8 x the off-diagonal cells makes the chain matrix indefinite."""
    from theiasfm_amd import lib
    prob = shared_problem(n_views=40, groups=(2, 9), seed=3)
    kw = dict(max_num_iterations=6)
    ref = oracle.solve(prob.copy(), options(abi.PRECOND_CLUSTER_TRIDIAGONAL, **kw))[1]
    monkeypatch.setenv("TMI_BA_TEST_TRI_SCALE0", "8.0")
    if path == "failure":
        monkeypatch.setenv("TMI_BA_TEST_TRI_SCALE1", "8.0")
    st_o, s_o = oracle.solve(prob.copy(), options(abi.PRECOND_CLUSTER_TRIDIAGONAL, **kw))
    st_d, s_d = lib.solve(prob.copy(), options(abi.PRECOND_CLUSTER_TRIDIAGONAL, **kw))
    assert s_d.num_iterations == s_o.num_iterations and s_d.num_successful_steps == s_o.num_successful_steps
    assert s_d.num_linear_solver_iterations == s_o.num_linear_solver_iterations
    if path == "retry":
        # the halved cells are a different (weaker) preconditioner than the unscaled ones: same minimum, another count
        assert st_d == st_o == 0 and s_d.success == s_o.success == 1
        assert abs(s_d.final_cost - s_o.final_cost) <= 1e-9 * s_o.final_cost
        assert s_o.num_linear_solver_iterations != ref.num_linear_solver_iterations
    else:
        # every linear solve fails: five invalid steps in a row end the solve (Ceres: max_num_consecutive_invalid_steps)
        assert s_d.success == s_o.success == 0 and s_d.num_successful_steps == 0 and s_d.num_linear_solver_iterations == 0
        assert abs(s_d.final_cost - s_d.initial_cost) <= 1e-12 * s_d.initial_cost  # nothing moved
        assert abs(s_o.final_cost - s_o.initial_cost) <= 1e-12 * s_o.initial_cost
        assert abs(s_d.final_cost - s_o.final_cost) <= 1e-12 * s_o.final_cost


@pytest.mark.gpu
def test_device_cluster_tridiagonal_without_the_formed_S_keeps_block_jacobi():
    """schur_mode implicit (and several ranks): no blocks between clusters to take -- the SCHUR_JACOBI blocks, and the
    summary says so"""
    from theiasfm_amd import lib
    prob = synth.config("ladybug49")
    kw = dict(max_num_iterations=6, schur_mode=abi.SCHUR_IMPLICIT)
    st_t, s_t = lib.solve(prob.copy(), options(abi.PRECOND_CLUSTER_TRIDIAGONAL, **kw))
    st_j, s_j = lib.solve(prob.copy(), options(abi.PRECOND_SCHUR_JACOBI, **kw))
    assert st_t == st_j == 0 and s_t.effective_preconditioner_type == abi.PRECOND_SCHUR_JACOBI
    assert s_t.num_linear_solver_iterations == s_j.num_linear_solver_iterations and s_t.final_cost == s_j.final_cost


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["mixed_sizes", "one_big_cluster", "pairs", "huber_dof4"])
def test_device_cluster_jacobi_matches_oracle(case):
    from theiasfm_amd import lib
    kw = {}
    if case == "mixed_sizes":
        prob = shared_problem()
    elif case == "one_big_cluster":
        prob = shared_problem(n_views=90, groups=(60, 60), seed=9, per_view=500)   # 60 views: 368 unknowns, 6 tile rows
    elif case == "pairs":
        prob = shared_problem(n_views=40, groups=(2, 2), seed=3)
    else:
        prob = shared_problem(n_views=36, groups=(3, 9), seed=11)
        kw = dict(loss_function_type=abi.LOSS_HUBER, robust_loss_width=3.0, point_dof=4)
    # the reference's stopping rules on (1e-6 / 1e-10 / 1e-8): iterations at the fixed point, where accepting a step is a
    # matter of round-off, would make the step counts of two correct implementations differ
    kw.update(function_tolerance=1e-6, gradient_tolerance=1e-10, parameter_tolerance=1e-8, max_num_iterations=12)
    o = options(abi.PRECOND_CLUSTER_JACOBI, **kw)
    a, b = prob.copy(), prob.copy()
    st_d, s_d = lib.solve(a, o)
    st_o, s_o = oracle.solve(b, o)
    assert st_d == st_o == 0, (s_d.message, s_o.message)
    assert s_d.num_iterations == s_o.num_iterations and s_d.num_successful_steps == s_o.num_successful_steps
    assert abs(int(s_d.num_linear_solver_iterations) - int(s_o.num_linear_solver_iterations)) <= 1
    assert abs(s_d.final_cost - s_o.final_cost) <= 1e-9 * s_o.final_cost
    assert abs(s_d.final_rmse - s_o.final_rmse) <= 1e-9
    assert np.abs(a.extrinsics - b.extrinsics).max() <= 1e-6 * 100.0
    assert np.abs(a.intrinsics - b.intrinsics).max() <= 1e-6 * max(1.0, np.abs(b.intrinsics).max())
    # and it is doing something: SCHUR_JACOBI on the same problem needs several times the PCG iterations
    c = prob.copy()
    st_j, s_j = lib.solve(c, options(abi.PRECOND_SCHUR_JACOBI, **kw))
    assert st_j == 0 and s_j.num_linear_solver_iterations > 2 * s_d.num_linear_solver_iterations


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [abi.SCHUR_IMPLICIT, abi.SCHUR_AUTO])
def test_device_cluster_jacobi_with_the_matrix_free_operator(mode):
    """schur_mode auto / implicit: the operator is matrix-free and only the blocks INSIDE the clusters are formed for the
    preconditioner to factor; same PCG, same trajectory as with the formed S."""
    from theiasfm_amd import lib
    prob = shared_problem()
    a, b, c = prob.copy(), prob.copy(), prob.copy()
    tol = dict(function_tolerance=1e-6, gradient_tolerance=1e-10, parameter_tolerance=1e-8, max_num_iterations=12)
    st_d, s_d = lib.solve(a, options(abi.PRECOND_CLUSTER_JACOBI, schur_mode=mode, **tol))
    st_e, s_e = lib.solve(c, options(abi.PRECOND_CLUSTER_JACOBI, schur_mode=abi.SCHUR_EXPLICIT, **tol))
    st_o, s_o = oracle.solve(b, options(abi.PRECOND_CLUSTER_JACOBI, **tol))
    assert st_d == st_e == st_o == 0, (s_d.message, s_o.message)
    assert s_d.num_matrix_free_iterations == s_d.num_iterations and s_e.num_matrix_free_iterations == 0
    assert 0 < s_d.num_schur_blocks < s_e.num_schur_blocks       # the clusters' blocks only
    assert s_d.num_iterations == s_o.num_iterations and s_d.num_successful_steps == s_o.num_successful_steps
    assert abs(int(s_d.num_linear_solver_iterations) - int(s_o.num_linear_solver_iterations)) <= 1
    assert abs(s_d.final_cost - s_o.final_cost) <= 1e-9 * s_o.final_cost
    assert abs(s_d.final_cost - s_e.final_cost) <= 1e-9 * s_e.final_cost
    assert np.abs(a.extrinsics - b.extrinsics).max() <= 1e-6 * 100.0


@pytest.mark.gpu
def test_device_cluster_jacobi_needs_its_blocks():
    """a handle created for SCHUR_JACOBI with the matrix-free operator has no blocks of S at all: asking that handle for
    CLUSTER_JACOBI at solve time falls back to SCHUR_JACOBI instead of failing"""
    from theiasfm_amd import lib
    prob = shared_problem(n_views=30, groups=(2, 8), seed=2)
    o_j = options(abi.PRECOND_SCHUR_JACOBI, schur_mode=abi.SCHUR_IMPLICIT, max_num_iterations=4)
    o_c = options(abi.PRECOND_CLUSTER_JACOBI, schur_mode=abi.SCHUR_IMPLICIT, max_num_iterations=4)
    sv = lib.Solver(prob.copy(), o_j)
    st1, s1 = sv.solve(o_j)
    sv.reset()
    st2, s2 = sv.solve(o_c)
    sv.close()
    assert st1 == st2 == 0
    assert s1.final_cost == s2.final_cost and s1.num_linear_solver_iterations == s2.num_linear_solver_iterations


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("mode", [abi.SCHUR_AUTO, abi.SCHUR_EXPLICIT])
def test_sharded_cluster_jacobi_matches_single_rank(world, mode):
    """tracks sharded over `world` ranks (threads on one device, the all-reduce hook sums their buffers): the clusters'
    blocks of S are partial sums on every rank and travel in the all-reduced system like the rest of it; every rank
    factors the same clusters and ends where the single-rank solve ends"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_sharded import _run_sharded
    from theiasfm_amd import lib
    prob = shared_problem(n_views=40, groups=(2, 12), seed=4)
    tol = dict(function_tolerance=1e-6, gradient_tolerance=1e-10, parameter_tolerance=1e-8, max_num_iterations=12)
    o = options(abi.PRECOND_CLUSTER_JACOBI, schur_mode=mode, **tol)
    single = prob.copy()
    st1, s1 = lib.solve(single, o)
    assert st1 == 0
    results = _run_sharded(prob, o, world)
    for st_r, s_r in results:
        assert st_r == 0 and s_r.success == 1
        assert s_r.num_iterations == s1.num_iterations and s_r.num_successful_steps == s1.num_successful_steps
        assert abs(int(s_r.num_linear_solver_iterations) - int(s1.num_linear_solver_iterations)) <= 1
        assert abs(s_r.final_cost - s1.final_cost) <= 1e-9 * s1.final_cost
        assert s_r.final_cost == results[0][1].final_cost

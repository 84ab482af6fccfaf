"""CPU tests of the host side of libsim3opt: the C-ABI surface, the graph container's error
behaviour (g2o's bool/int conventions turned into status codes), the reference-format loader and
the row partition.  No compute call is made here -- without a GPU the library must fail loudly.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

from sim3opt_amd import lib as L, sim3np as S3
import kitti_graph as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
I8 = [0, 0, 0, 1, 0, 0, 0, 1.0]


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "sim3opt.h")).read()
    bench = open(os.path.join(ROOT, "include", "sim3opt_bench.h")).read()
    # (what only a SIM3OPT_BENCH_HOOKS build exports -- measurement prototypes -- is not part of the product)
    bench = re.sub(r"#ifdef SIM3OPT_BENCH_HOOKS.*?#endif", "", bench, flags=re.S)
    # the drop-in interface carries no measurement hooks (they live in sim3opt_bench.h)
    assert "sim3opt_bench_" not in hdr and set(re.findall(r"\b(sim3opt_[a-z0-9_]+)\s*\(", bench)) == {
        "sim3opt_bench_spmv", "sim3opt_bench_stream"}
    declared = set(re.findall(r"\b(sim3opt_[a-z0-9_]+)\s*\(", hdr + bench))
    declared -= {"sim3opt_graph", "sim3opt_options", "sim3opt_iter_stats", "sim3opt_kernel_times", "sim3opt_alltoallv_fn"}
    assert declared == set(L.SYMBOLS), declared ^ set(L.SYMBOLS)
    lib = L.load()  # binds every symbol, raises AttributeError on a missing export
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.sim3opt_version() >= 100


def test_options_struct_matches_header_defaults():
    o = L.default_options()
    assert o.tau == 1e-5 and o.max_trials == 10 and o.fd_delta == 1e-9 and o.exp_eps == 1e-5
    assert abs(o.good_step_lower - 1 / 3) < 1e-16 and abs(o.good_step_upper - 2 / 3) < 1e-16
    assert o.fix_small_angle_b == 0 and o.device == -1 and o.dof_mask == 127 and o.pcg_graph == 1
    # field-by-field agreement between the ctypes mirror and the C header
    hdr = open(os.path.join(ROOT, "include", "sim3opt.h")).read()
    body = hdr[hdr.index("typedef struct sim3opt_options {"):hdr.index("} sim3opt_options;")]
    fields = re.findall(r"^\s*(?:double|int32_t|int64_t)\s+([a-z_0-9]+)(?:\[\d+\])?;", body, flags=re.M)
    assert fields == [f[0] for f in L.Options._fields_]
    # the numeric tuning knobs are options (an environment variable of the same name is a debug override that
    # sim3opt_get_options echoes after sim3opt_initialize)
    assert list(o.amg_cycle) == [0, 0, 0, 0] and o.amg_fp32 == 1 and o.amg_omega == 0.9 and o.adaptive_prec == 1
    assert list(o.amg_over) == [1.8, 1.6] and o.amg_coarsest == 256 and o.row_order == -1 and o.halo_exchange == 1
    G = L.Graph(pcg_rel_tol=1e-6, verbose=1)
    assert G.options().pcg_rel_tol == 1e-6 and G.options().verbose == 1
    with pytest.raises(L.Sim3OptError):
        G.set_options(fd_delta=0.0)


def test_graph_container_error_codes():
    G = L.Graph()
    G.add_vertex(10, I8, fixed=True)
    G.add_vertex(-7, I8)
    with pytest.raises(L.Sim3OptError) as ei:  # duplicate id: g2o's addVertex returns false
        G.add_vertex(10, I8)
    assert ei.value.code == L.ERR_ARG
    with pytest.raises(L.Sim3OptError):  # unknown endpoint
        G.add_edge(10, 99, I8)
    with pytest.raises(L.Sim3OptError):  # identical endpoints
        G.add_edge(10, 10, I8)
    with pytest.raises(L.Sim3OptError):  # non-positive scale
        G.add_edge(10, -7, [0, 0, 0, 1, 0, 0, 0, 0.0])
    with pytest.raises(L.Sim3OptError):  # NaN state
        G.add_vertex(3, [np.nan, 0, 0, 1, 0, 0, 0, 1])
    with pytest.raises(L.Sim3OptError):  # Huber needs delta > 0
        G.add_edge(10, -7, I8, kernel=L.KERNEL_HUBER, kernel_delta=0.0)
    G.add_edge(10, -7, I8)
    assert (G.num_vertices, G.num_edges) == (2, 1)
    a, b, m = G.get_edge(0)
    assert (a, b) == (10, -7) and np.array_equal(m, I8)
    # estimates can be read and warm-started before initialize (vertex objects in g2o)
    s = np.array([0, 0, 0, 1, 1, 2, 3, 1.5])
    G.set_vertex(-7, s)
    assert np.array_equal(G.get_vertex(-7), s)
    with pytest.raises(L.Sim3OptError):
        G.get_vertex(12345)
    # compute entry points refuse to run before initialize
    for fn in (G.chi2, G.linearize, G.edge_errors):
        with pytest.raises(L.Sim3OptError) as ei:
            fn()
        assert ei.value.code == L.ERR_STATE


def test_no_cpu_fallback_without_gpu():
    """On a machine without a HIP device initialize() must fail loudly, never fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    G = L.Graph()
    G.add_vertices(np.tile(I8, (3, 1)), [1, 0, 0])
    G.add_edges([1, 2], [0, 1], np.tile(I8, (2, 1)))
    with pytest.raises(L.Sim3OptError) as ei:
        G.initialize()
    assert ei.value.code == L.ERR_NO_DEVICE
    with pytest.raises(L.Sim3OptError):
        G.optimize(5)


def test_tuning_knobs_are_options_and_environment_overrides_are_echoed(monkeypatch):
    """The numeric tuning of the PCG path lives in sim3opt_options; a SIM3OPT_* environment variable is a debug
    override applied by sim3opt_initialize and reported by sim3opt_get_options (host logic: works without a GPU --
    the override is applied before the device is asked for)."""
    G = L.Graph(amg_cycle=[1, 2, 2, 2], amg_over=[1.7, 1.5], amg_shard_rows=2000, pcg_batch=3)
    o = G.options()
    assert list(o.amg_cycle) == [1, 2, 2, 2] and list(o.amg_over) == [1.7, 1.5]
    assert o.amg_shard_rows == 2000 and o.pcg_batch == 3
    with pytest.raises(L.Sim3OptError):
        G.set_options(amg_omega=0.0)
    monkeypatch.setenv("SIM3OPT_AMG_CYCLE", "23")
    monkeypatch.setenv("SIM3OPT_AMG_OVER", "1.9,1.7")
    monkeypatch.setenv("SIM3OPT_ROW_ORDER", "bfs")
    monkeypatch.setenv("SIM3OPT_NO_HALO", "1")
    monkeypatch.setenv("SIM3OPT_PCG_BATCH", "1")
    G.add_vertices(np.tile(I8, (3, 1)), [1, 0, 0])
    G.add_edges([1, 2], [0, 1], np.tile(I8, (2, 1)))
    try:
        G.initialize()
    except L.Sim3OptError as e:  # no GPU here: the overrides were applied before the device was asked for
        assert e.code == L.ERR_NO_DEVICE
    o = G.options()
    assert list(o.amg_cycle) == [2, 3, 3, 3] and list(o.amg_over) == [1.9, 1.7]
    assert o.row_order == 1 and o.halo_exchange == 0 and o.pcg_batch == 1
    assert o.amg_shard_rows == 2000  # (no variable set: the caller's value stays)


def test_partition_aware_hierarchy_and_halo_plan_host_side():
    """Host parts of the multi-rank design (no GPU): with amg_virtual_ranks = N the aggregates stay inside the N
    equal row spans (so Galerkin rows and restrictions need no reduction across ranks) and every rank's
    neighbour plan mirrors its neighbours'."""
    from sim3opt_amd import synth
    synth.DRIFT_TARGET = 0.05
    g = synth.manhattan(1500, 15000, dims=(12, 12, 10))
    for world in (2, 4, 8):
        G = L.Graph(row_order=1, amg_virtual_ranks=world, amg_coarsest=64)
        G.add_vertices(g["states"], g["fixed"])
        G.add_edges(g["v0"], g["v1"], g["meas"])
        rows, blocks, agg = G.amg_hierarchy()
        assert len(rows) >= 3 and rows[0] == 1499
        _, beg, bnd, _ = G.partition_plan(world)
        first = 0
        for r in range(world):
            a = np.unique(agg[beg[r]:beg[r + 1]])
            assert a[0] == first and a[-1] - a[0] + 1 == a.size  # contiguous, nothing shared with another rank
            first = a[-1] + 1
        assert first == rows[1]
        plans = [G.halo_plan(world, r) for r in range(world)]
        for p in range(world):
            sp, ssp, rp, rsp = plans[p]
            assert rsp[p + 1] == rsp[p] and ssp[p + 1] == ssp[p]  # nothing to itself
            assert np.all((rp < beg[p]) | (rp >= beg[p + 1])) and np.all((sp >= beg[p]) & (sp < beg[p + 1]))
            for q in range(world):
                sq, ssq, rq, rsq = plans[q]
                assert np.array_equal(rp[rsp[q]:rsp[q + 1]], sq[ssq[p]:ssq[p + 1]])
            # what a rank sends is its boundary: the rows with a neighbour on another rank
            assert len(np.unique(sp)) == bnd[p]
        # the partition only forbids pairs across span borders: the hierarchy stays about as coarse as the
        # unpartitioned one (the greedy matching is not monotone, hence "about")
        G1 = L.Graph(row_order=1, amg_coarsest=64)
        G1.add_vertices(g["states"], g["fixed"])
        G1.add_edges(g["v0"], g["v1"], g["meas"])
        assert abs(int(G1.amg_hierarchy()[0][1]) - int(rows[1])) <= 0.15 * rows[1]


def test_nothing_to_optimise_conventions():
    G = L.Graph()
    assert G._L.sim3opt_optimize(G._g, 10) == -1  # empty graph: g2o returns -1
    G.add_vertices(np.tile(I8, (2, 1)), [1, 1])
    G.add_edges([1], [0], [I8])
    with pytest.raises(L.Sim3OptError) as ei:  # all vertices fixed
        G.initialize()
    assert ei.value.code in (L.ERR_STATE, L.ERR_NO_DEVICE)


@pytest.mark.parametrize("one", [True, False])
def test_kitti_loader_matches_python_builder(one):
    """C++ loader (kitti_io.cpp, mirrors kitti_surf.cpp:145-292, 575-670) vs the numpy builder."""
    ref = K.build_direct_graph(one)
    G = L.Graph()
    G.load_kitti_direct(K.FIXTURE, one)
    assert G.num_vertices == 771 and G.num_edges == (771 if one else 888)
    st = G.get_vertices()
    assert np.abs(st - ref["states"]).max() < 1e-12
    for k in list(range(0, G.num_edges, 37)) + [G.num_edges - 1]:
        a, b, m = G.get_edge(k)
        assert (a, b) == (ref["v0"][k], ref["v1"][k])
        assert np.abs(m - ref["meas"][k]).max() < 1e-12
    with pytest.raises(L.Sim3OptError) as ei:
        L.Graph().load_kitti_direct("/nonexistent/dir", True)
    assert ei.value.code == L.ERR_IO


def test_write_poses_format(tmp_path):
    G = L.Graph()
    G.load_kitti_direct(K.FIXTURE, True)
    path = str(tmp_path / "direct_pure.txt")
    G.write_poses(path, K.load_cc())
    lines = open(path).read().splitlines()
    assert lines[0].startswith("%") and len(lines) == 772
    row = [float(x) for x in lines[6].split()]
    st = G.get_vertices()[5]
    Swi = S3.inv(st)
    assert int(row[0]) == K.load_cc()[5] and row[1] == st[7]
    assert np.abs(np.array(row[2:5]) - Swi[4:7]).max() < 1e-15
    assert np.abs(np.array(row[5:9]) - Swi[:4]).max() < 1e-15


def test_partition_rows_balances_blocks():
    rng = np.random.default_rng(0)
    deg = rng.integers(1, 30, size=1000)
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    for world in (1, 2, 3, 8):
        beg = L.partition_rows(rowptr, world)
        assert beg[0] == 0 and beg[-1] == 1000 and np.all(np.diff(beg) >= 0)
        loads = [rowptr[beg[r + 1]] - rowptr[beg[r]] for r in range(world)]
        assert max(loads) - min(loads) <= 2 * deg.max()
    beg = L.partition_rows(np.array([0, 5], dtype=np.int32), 4)  # fewer rows than ranks
    assert beg[0] == 0 and beg[-1] == 1 and np.all(np.diff(beg) >= 0)


def test_locality_order_and_halo_of_the_row_partition():
    """The partitioned path numbers the block rows breadth-first from a pseudo-peripheral vertex, so
    that contiguous rank spans are slabs of the graph: on a Manhattan world a minority of the rows
    has a neighbour on another rank (what the per-iteration halo exchange sends) and few edges are
    cut (what two ranks linearise) -- in insertion order, along the random walk, nearly all are."""
    from sim3opt_amd import synth
    g = synth.manhattan(20000, 200000, dims=(40, 40, 10))
    G = L.Graph()
    G.add_vertices(g["states"], g["fixed"])
    G.add_edges(g["v0"], g["v1"], g["meas"])
    nb = 19999
    for world in (2, 4, 8):
        v, rb, bnd, cut = G.partition_plan(world, locality=True)
        v0, rb0, bnd0, cut0 = G.partition_plan(world, locality=False)
        assert sorted(v) == sorted(v0) == list(np.where(g["fixed"] == 0)[0])  # a permutation of the free vertices
        assert np.array_equal(v0, np.where(g["fixed"] == 0)[0])                 # insertion order is g2o's
        assert np.array_equal(rb, rb0) and rb[0] == 0 and rb[-1] == nb
        assert bnd.sum() < 0.6 * bnd0.sum() and cut < 0.35 * cut0, (world, bnd.sum(), bnd0.sum(), cut, cut0)
        # the boundary list is exactly the rows with a cross-rank neighbour
        row_of = -np.ones(len(g["fixed"]), dtype=np.int64)
        row_of[v] = np.arange(nb)
        a, b = row_of[g["v0"]], row_of[g["v1"]]
        ok = (a >= 0) & (b >= 0)
        ra, rbk = np.searchsorted(rb, a[ok], side="right") - 1, np.searchsorted(rb, b[ok], side="right") - 1
        crossing = ra != rbk
        assert cut == int(crossing.sum())
        rows = np.unique(np.concatenate([a[ok][crossing], b[ok][crossing]]))
        assert [int(((rows >= rb[r]) & (rows < rb[r + 1])).sum()) for r in range(world)] == list(bnd)
    # one rank: nothing to exchange
    assert G.partition_plan(1)[2].sum() == 0 and G.partition_plan(1)[3] == 0
    # a graph in two components is ordered component by component
    H = L.Graph()
    H.add_vertices(np.tile(I8, (7, 1)), [1, 0, 0, 0, 0, 0, 0])
    H.add_edges([0, 1, 4, 5], [1, 2, 5, 6], np.tile(I8, (4, 1)))
    v = H.partition_plan(2)[0]
    assert sorted(v) == [1, 2, 3, 4, 5, 6] and set(v[:2]) == {1, 2} or set(v[-2:]) == {1, 2}


def test_allgather_plan_for_uneven_and_empty_ranks():
    """The multi-GPU exchange is ONE in-place equal-count ncclAllGather: the library's own predicate
    and buffer sizing (comm.hpp allgather_equal_plan, used by the RCCL transport and by the engine's
    allocations), for row counts that do not divide by the rank count and for ranks left empty."""
    import ctypes as C
    lib = L.load()
    for nb in (1, 2, 3, 7, 8, 9, 63, 64, 770, 99999, 100000):
        for world in (1, 2, 3, 4, 5, 8):
            rb = np.zeros(world + 1, dtype=np.int32)
            cnt, padded = C.c_int64(), C.c_int64()
            rc = lib.sim3opt_comm_allgather_plan(nb, world, rb.ctypes.data_as(L._ip), C.byref(cnt),
                                                 C.byref(padded))
            assert rc == 1, (nb, world)  # the equal-count path always applies to this partition
            rpr = -(-nb // world)
            assert rb[0] == 0 and rb[-1] == nb and np.all(np.diff(rb) >= 0)
            assert all(int(rb[r + 1] - rb[r]) in (rpr, nb - min(r * rpr, nb)) or rb[r + 1] == rb[r]
                       for r in range(world))
            assert cnt.value == 7 * rpr and padded.value == 7 * rpr * world >= 7 * nb
            # rank r's segment sits at r * count inside the padded buffer; empty trailing ranks own
            # padding only, and no owned segment reaches into another rank's slot
            for r in range(world):
                lo, hi = 7 * int(rb[r]), 7 * int(rb[r + 1])
                assert lo == min(r * cnt.value, 7 * nb) and hi - lo <= cnt.value
    assert lib.sim3opt_comm_allgather_plan(-1, 2, None, None, None) < 0


def test_umeyama_alignment_and_kitti_original_map_rmse():
    """Evaluation harness (kitti_surf.cpp:1091-1161, :1427-1463): C++ Umeyama vs numpy SVD, and
    the RMSE of the un-optimised VO keyframe trajectory against KITTI-00 ground truth."""
    def umeyama_np(x, y):
        mx, my = x.mean(0), y.mean(0)
        dx, dy = x - mx, y - my
        U, D, Vt = np.linalg.svd(dy.T @ dx / len(x))
        Sg = np.eye(3)
        if np.linalg.det(U) * np.linalg.det(Vt) < 0:
            Sg[2, 2] = -1
        R = U @ Sg @ Vt
        c = np.trace(np.diag(D) @ Sg) / (dx ** 2).sum(1).mean()
        M = np.eye(4)
        M[:3, :3] = c * R
        M[:3, 3] = my - c * R @ mx
        return M

    rng = np.random.default_rng(0)
    for trial in range(5):
        x = rng.standard_normal((40, 3)) * 5
        if trial == 3:
            x[:, 2] = 0.0  # coplanar points: the third singular value vanishes
        q = rng.standard_normal(4)
        R = S3.quat_to_R(q / np.linalg.norm(q))
        if trial == 4:
            R = R @ np.diag([1, 1, -1.0])  # a reflection in the data must not leak into R
        y = 2.5 * (x @ R.T) + np.array([1, 2, 3]) + rng.standard_normal((40, 3)) * 0.01
        S, rm, mx = L.align_trajectory(x, y)
        assert np.abs(S - umeyama_np(x, y)).max() < 1e-9
        assert abs(np.linalg.det(S[:3, :3] / np.cbrt(np.linalg.det(S[:3, :3]))) - 1) < 1e-9
        al = x @ S[:3, :3].T + S[:3, 3]
        assert abs(rm - np.sqrt(((y - al) ** 2).sum(1).mean())) < 1e-12
        assert abs(mx - np.sqrt(((y - al) ** 2).sum(1)).max()) < 1e-12
    gt = np.loadtxt(os.path.join(K.FIXTURE, "gt_kf.txt"), comments="%")
    assert gt.shape == (771, 13) and list(gt[:, 0].astype(int)) == K.load_cc()
    pos = S3.inv(K.build_direct_graph(True)["states"])[:, 4:7]
    S, rm, mx = L.align_trajectory(pos, gt[:, [4, 8, 12]])
    assert abs(rm - 130.236) < 1e-2 and abs(mx - 264.069) < 1e-2  # un-optimised VO map vs GT
    with pytest.raises(L.Sim3OptError):
        L.align_trajectory(np.zeros((2, 3)), np.zeros((2, 3)))


def test_amg_hierarchy_structure():
    """Host set-up of the multigrid preconditioner (amg.cpp): pairwise matching coarsens a
    Manhattan graph by ~8x per level down to <= 256 rows; aggregates are connected subsets of 8 block
    rows (a few larger ones where left-over rows joined a neighbour); a star graph (nothing to match after the hub) is refused."""
    from sim3opt_amd import synth
    g = synth.manhattan(3000, 30000, dims=(17, 17, 10))
    G = L.Graph()
    G.add_vertices(g["states"], g["fixed"])
    G.add_edges(g["v0"], g["v1"], g["meas"])
    rows, blocks, agg = G.amg_hierarchy()
    assert rows[0] == 2999 and blocks[0] == 2999 + 2 * 30000 - 2 * int(((g["v0"] == 0) | (g["v1"] == 0)).sum())
    assert len(rows) >= 3 and rows[-1] <= 256
    assert all(rows[l + 1] <= rows[l] // 3 for l in range(len(rows) - 2))  # 3 matching passes per level
    assert all(blocks[l + 1] < blocks[l] for l in range(len(rows) - 1))
    agg = agg[:2999]
    assert agg.min() == 0 and agg.max() == rows[1] - 1
    sizes = np.bincount(agg)
    assert sizes.max() <= 64 and sizes.min() >= 1 and np.median(sizes) == 8
    # every aggregate is connected in the graph (matching only ever merges neighbours)
    free_of = np.cumsum(g["fixed"] == 0) - 1
    a, b = free_of[g["v0"]], free_of[g["v1"]]
    ok = (g["fixed"][g["v0"]] == 0) & (g["fixed"][g["v1"]] == 0)
    a, b = a[ok], b[ok]
    same = agg[a] == agg[b]
    import scipy.sparse as sp
    import scipy.sparse.csgraph as csg
    A = sp.coo_matrix((np.ones(same.sum()), (a[same], b[same])), shape=(2999, 2999))
    ncomp, _ = csg.connected_components(A, directed=False)
    assert ncomp == rows[1]
    # deterministic
    rows2, blocks2, agg2 = G.amg_hierarchy()
    assert np.array_equal(rows, rows2) and np.array_equal(agg[:2999], agg2[:2999])
    # star: vertex 1 tied to everyone, nothing else -> after the first pair nothing matches
    S = L.Graph()
    n = 400
    S.add_vertices(np.tile(I8, (n, 1)), [1] + [0] * (n - 1))
    S.add_edges(np.full(n - 2, 1), np.arange(2, n), np.tile(I8, (n - 2, 1)))
    with pytest.raises(L.Sim3OptError) as ei:
        S.amg_hierarchy()
    assert ei.value.code == L.ERR_STATE


def test_amg_hierarchy_on_the_kitti_fixture():
    """The 771-keyframe chain coarsens to a two-level hierarchy with aggregates of 4 consecutive
    keyframes (770 -> 192 rows, block-tridiagonal coarse pattern); the 117 extra loop edges change
    a few aggregates, not the shape.  Pins the aggregation against silent changes."""
    G = L.Graph()
    G.load_kitti_direct(K.FIXTURE, True)
    rows, blocks, agg = G.amg_hierarchy()
    assert list(rows) == [770, 192] and list(blocks) == [2310, 576]
    assert list(agg[:770]) == [i // 4 for i in range(768)] + [191, 191]
    G = L.Graph()
    G.load_kitti_direct(K.FIXTURE, False)
    rows, blocks, agg = G.amg_hierarchy()
    assert list(rows) == [770, 191] and list(blocks) == [2544, 641]
    assert np.bincount(agg[:770]).max() <= 8

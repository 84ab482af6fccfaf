"""Host-side plan of the exact sparse block Cholesky (sim3opt_amd/csrc/direct.cpp), no GPU needed.

The plan is executed here in numpy exactly as the device kernels walk it (groups, levels, phases;
direct_kernels.hpp) on a random SPD system with the graph's block pattern and compared with a dense
solve: this checks the nested-dissection order, the symbolic factorisation, the source lists, the
update lists and the schedule's dependency order without a GPU.  The role it fills in the reference:
LinearSolverEigen / SimplicialLDLT (kitti_surf.cpp:553-554)."""
import numpy as np
import pytest

from sim3opt_amd import lib as L, synth
import kitti_graph as K


def run_plan(P, vals, b, lam):
    nb, nL = P["nb"], P["nL"]
    Lb = np.zeros((nL, 7, 7))
    y = np.zeros((nb, 7))
    x = np.zeros((nb, 7))
    xo = np.zeros((nb, 7))
    lcol = np.repeat(np.arange(nb), np.diff(P["colptr"]))
    done = np.zeros(nb, bool)
    for g in range(P["ngroups"]):
        for l in range(P["gptr"][g], P["gptr"][g + 1]):
            c0, c1 = P["lcolp"][l], P["lcolp"][l + 1]
            for s in range(P["colptr"][c0], P["colptr"][c1]):  # phase A
                acc = np.zeros((7, 7))
                for k in range(P["srcptr"][s], P["srcptr"][s + 1]):
                    acc += vals[P["src"][k]]
                if s == P["colptr"][lcol[s]]:
                    acc += lam * np.eye(7)
                for k in range(P["pairptr"][s], P["pairptr"][s + 1]):
                    sa, sb = P["pa"][k], P["pb"][k]
                    assert done[lcol[sa]] and done[lcol[sb]], "schedule violates a dependency"
                    acc -= Lb[sa] @ Lb[sb].T
                Lb[s] = acc
            for j in range(c0, c1):  # phases B and C
                s0 = P["colptr"][j]
                Ljj = np.linalg.cholesky(Lb[s0])
                Lb[s0] = Ljj
                t = b[P["perm"][j]].copy()
                for k in range(P["pairptr"][s0], P["pairptr"][s0 + 1]):
                    sa = P["pa"][k]
                    t -= Lb[sa] @ y[lcol[sa]]
                y[j] = np.linalg.solve(Ljj, t)
                for s in range(s0 + 1, P["colptr"][j + 1]):
                    Lb[s] = np.linalg.solve(Ljj, Lb[s].T).T
            done[c0:c1] = True
    for g in reversed(range(P["ngroups"])):  # the top group first, then the subtrees
        for l in range(P["gptr"][g + 1] - 1, P["gptr"][g] - 1, -1):
            for j in range(P["lcolp"][l], P["lcolp"][l + 1]):
                s0 = P["colptr"][j]
                t = y[j].copy()
                for s in range(s0 + 1, P["colptr"][j + 1]):
                    t -= Lb[s].T @ x[P["lrow"][s]]
                x[j] = np.linalg.solve(Lb[s0].T, t)
                xo[P["perm"][j]] = x[j]
    return xo


def random_spd_on_pattern(rowptr, colidx, seed):
    """Block values of a sum of edge terms [Ja Jc]^T [Ja Jc] on the graph's own pattern (parallel edges
    keep separate blocks, like the linearisation kernel's output)."""
    rng = np.random.default_rng(seed)
    nb, nnzb = len(rowptr) - 1, len(colidx)
    rows = np.repeat(np.arange(nb), np.diff(rowptr))
    vals = np.zeros((nnzb, 7, 7))
    seen = {}
    for k in range(nnzb):
        if rows[k] != colidx[k]:
            seen.setdefault((rows[k], colidx[k]), []).append(k)
    for (a, c), ks in seen.items():
        if a < c:
            for k, k2 in zip(ks, seen[(c, a)]):
                Ja, Jc = rng.standard_normal((7, 7)), rng.standard_normal((7, 7))
                vals[rowptr[a]] += Ja.T @ Ja
                vals[rowptr[c]] += Jc.T @ Jc
                vals[k] = Ja.T @ Jc
                vals[k2] = Jc.T @ Ja
    M = np.zeros((7 * nb, 7 * nb))
    for k in range(nnzb):
        a, c = rows[k], colidx[k]
        M[7 * a:7 * a + 7, 7 * c:7 * c + 7] += vals[k]
    return vals, M


def graph_of(g):
    G = L.Graph()
    G.add_vertices(g["states"], g["fixed"])
    G.add_edges(g["v0"], g["v1"], g["meas"])
    return G


CASES = {
    "kitti_one_loop": lambda: K.build_direct_graph(True),
    "kitti_all_loops": lambda: K.build_direct_graph(False),
    "manhattan_300": lambda: synth.manhattan(300, 1500, dims=(8, 8, 3)),
    "chain_200": lambda: synth.chain_loop(200, 230),
    "tiny_5": lambda: synth.chain_loop(5, 6, min_gap=2),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_plan_solves_like_dense(name):
    G = graph_of(CASES[name]())
    P = G.direct_plan()
    rowptr, colidx = G.system_pattern()
    nb = P["nb"]
    assert nb == len(rowptr) - 1 and sorted(P["perm"]) == list(range(nb))
    assert P["colptr"][-1] == P["nL"] and P["pairptr"][-1] == P["npairs"]
    assert P["lcolp"][0] == 0 and P["lcolp"][-1] == nb and P["gptr"][-1] == P["nlevels"]
    for j in range(nb):  # diagonal first, rows ascending
        r = P["lrow"][P["colptr"][j]:P["colptr"][j + 1]]
        assert r[0] == j and (np.diff(r) > 0).all()
    # the kernel's work split: the rounds of a level tile its blocks exactly once, in order, in cells
    # of at most 8 blocks, and every cell knows its product range
    assert len(P["rptr"]) == P["nlevels"] + 1 and P["rptr"][-1] == P["nrounds"]
    for g in range(P["ngroups"]):
        nw = 8  # (bottom groups run 8 wavefronts too since round 3: DirectPlan::sub_waves)
        for l in range(P["gptr"][g], P["gptr"][g + 1]):
            S0, S1 = P["colptr"][P["lcolp"][l]], P["colptr"][P["lcolp"][l + 1]]
            nxt = S0
            for q in range(P["rptr"][l], P["rptr"][l + 1]):
                cell = P["cells"][18 * q:18 * q + 18]
                assert cell[0] == nxt and (np.diff(cell[:9]) >= 0).all() and (np.diff(cell[:nw + 1]) <= 8).all()
                assert (cell[nw:9] == cell[nw]).all()  # wavefronts beyond the group's own stay idle
                assert (cell[9:] == P["pairptr"][cell[:9]]).all()
                nxt = cell[nw]
            assert nxt == S1 and P["rptr"][l + 1] > P["rptr"][l]
    vals, M = random_spd_on_pattern(rowptr, colidx, 11)
    b = np.random.default_rng(12).standard_normal((nb, 7))
    lam = 0.5
    xref = np.linalg.solve(M + lam * np.eye(7 * nb), b.ravel()).reshape(nb, 7)
    x = run_plan(P, vals, b, lam)
    assert np.abs(x - xref).max() < 1e-11 * np.abs(xref).max()


def test_kitti_plan_is_shallow_and_sparse():
    """What makes the factorisation GPU-friendly: nested dissection keeps the elimination tree of the
    770-keyframe chain 10-20 levels tall (a minimum-degree order: 770) with almost no fill."""
    for one, max_h, max_blocks in ((True, 16, 2400), (False, 26, 3000)):
        P = graph_of(K.build_direct_graph(one)).direct_plan()
        assert P["nb"] == 770 and P["height"] <= max_h and P["nL"] <= max_blocks
        assert P["ngroups"] > 8  # independent bottom subtrees + the top


def test_plan_refuses_graphs_with_heavy_fill():
    """A 3-D Manhattan world fills in: the automatic limit hands it to the PCG."""
    G = graph_of(synth.manhattan(1000, 10000, dims=(10, 10, 5)))
    with pytest.raises(L.Sim3OptError):
        G.direct_plan()
    P = G.direct_plan(max_pairs=50_000_000)  # ... unless the caller insists
    assert P["npairs"] > 300_000

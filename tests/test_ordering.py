"""Host-side ordering of the triangular factors (no GPU): multicolour permutation with LINE GROUPS — pairs of velocity
nodes / triples of pressure DoFs that follow each other on a lattice line, coloured as one vertex — through the
library's own analysis code (nsk_debug_tri_ordering), and what it does to the reference algorithm's inner iteration
counts (CPU oracle with the same permutations)."""
import numpy as np
import pytest
import scipy.sparse as sp
from types import SimpleNamespace

from navier_stokes_solver_amd import solver as S
from tests.util import problem


def _schur_block(pr):
    Sm = (pr.B.to_scipy() @ sp.diags(1.0 / pr.F.to_scipy().diagonal()) @ pr.Bt.to_scipy()).tocsr()
    Sm.sort_indices()
    return SimpleNamespace(rows=pr.n_p, rowptr=Sm.indptr, col=Sm.indices), Sm


def _check(perm, chain, A, xy, items_are_nodes):
    n = A.shape[0]
    assert sorted(perm.tolist()) == list(range(n))
    step = 2 if items_are_nodes else 1
    items = perm[::step] // step
    pos, length = chain & 15, chain >> 4
    assert len(items) == len(chain) and np.all(pos < length)
    # members of a group: consecutive permuted items, same y, ascending x, each adjacent to the one before
    A = sp.csr_matrix((np.ones(len(A.data)), A.indices, A.indptr), shape=A.shape)     # the PATTERN (stored zeros count)
    G = (A + A.T).tocsr()
    if items_are_nodes:
        Q = sp.csr_matrix((np.ones(n), (np.arange(n) // 2, np.arange(n))), shape=(n // 2, n))
        G = (Q @ G @ Q.T).tocsr()
    pts = xy[::step]
    follow = np.nonzero(pos > 0)[0]
    a, b = items[follow - 1], items[follow]
    assert np.all(pos[follow - 1] == pos[follow] - 1)
    assert np.allclose(pts[a, 1], pts[b, 1], rtol=0, atol=1e-12) and np.all(pts[b, 0] > pts[a, 0])
    assert np.all(np.asarray(G[a, b]).ravel() != 0)
    # colours: the groups of one colour are pairwise non-adjacent.  A colour starts where the first members' ids fall.
    gid = np.cumsum(pos == 0) - 1
    first_of_group = items[pos == 0]
    colour_of_group = np.concatenate([[0], np.cumsum(np.diff(first_of_group) < 0)])
    colour = colour_of_group[gid]
    item_colour = np.empty(len(items), int)
    item_group = np.empty(len(items), int)
    item_colour[items], item_group[items] = colour, gid
    C = G.tocoo()
    clash = (item_colour[C.row] == item_colour[C.col]) & (item_group[C.row] != item_group[C.col])
    assert not clash.any()
    return int(colour.max()) + 1


@pytest.mark.parametrize("name", ["ns16", "ns60"])
def test_line_group_ordering_is_a_valid_multicolouring(name):
    pr = problem(name)
    Sblk, Sm = _schur_block(pr)
    pF, kF, gF, b2, chF = S.tri_ordering_host(pr.F, pr.support_u, 2, want_block2=True)
    pS, kS, gS, _, chS = S.tri_ordering_host(Sblk, pr.support_p, 3)
    assert b2 and gF == 2 and gS == 3
    assert _check(pF, chF, pr.F.to_scipy(), pr.support_u, True) == kF
    assert _check(pS, chS, Sm, pr.support_p, False) == kS
    p1, k1, g1, _, ch1 = S.tri_ordering_host(pr.F, None, 2, want_block2=True)      # no support points: plain colouring
    pS1, kS1, gS1, _, _ = S.tri_ordering_host(Sblk, None, 3)
    assert g1 == 1 and gS1 == 1 and np.all(ch1 == 16)
    if name == "ns60":
        assert kF <= 13 < k1 and kS <= 19 < kS1                 # 12 instead of 17 node colours, 17-18 instead of 29-31
    # sub-domains: no group crosses a cut
    off = np.array([0, (pr.n_p // 3), 2 * (pr.n_p // 3), pr.n_p], np.int32)
    pS3, _, _, _, ch3 = S.tri_ordering_host(Sblk, pr.support_p, 3, sub_off=off)
    follow = np.nonzero((ch3 & 15) > 0)[0]
    shard = np.searchsorted(off, pS3, side="right") - 1
    assert np.all(shard[follow] == shard[follow - 1])


def test_line_groups_keep_the_inner_iteration_counts():
    """FGMRES + aSIMPLE on the 60x20 Newton system, ten outer iterations with the CPU oracle: the inner iteration counts
    with the line-group orderings stay within a few per cent of the one-DoF-at-a-time colouring's (the larger meshes of
    the study in DESIGN.md: F 20.7 vs 21.9, S 59.6 vs 58.9 per application at 120x40)."""
    from oracle import oracle as O
    pr = problem("ns60")
    Sblk, _ = _schur_block(pr)
    b, x0 = np.concatenate([pr.rhs_u, pr.rhs_p]), np.concatenate([pr.x0_u, pr.x0_p])
    its = {}
    for gu, gp in ((1, 1), (2, 3)):
        pF = S.tri_ordering_host(pr.F, pr.support_u, gu, want_block2=True)[0]
        pS = S.tri_ordering_host(Sblk, pr.support_p, gp)[0]
        _, info = O.OracleProblem.from_local(pr, perm_F=pF, perm_S=pS).solve(b, x0, solver=1, prec=2, variant=0, tol=0.0, max_iter=10)
        its[(gu, gp)] = (info["inner_u_its"] / info["prec_applies"], info["inner_p_its"] / info["prec_applies"])
    (f1, s1), (f2, s2) = its[(1, 1)], its[(2, 3)]
    assert f2 <= 1.15 * f1 and s2 <= 1.05 * s1, its

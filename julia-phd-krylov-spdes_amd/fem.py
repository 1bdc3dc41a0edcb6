"""Host-side set-up that feeds the Schur-PCG hot path (boundary producer).

In the reference all of this runs once per realization on the host (Julia) and hands
arrays-of-SparseMatrixCSC plus index maps to the operators. It is NOT the hot path and
has no device kernels here either; it exists so that tests, `bench.py` and `smoke()` have
blocks of exactly the reference's shape to feed through the C ABI.

Citations are relative to /root/reference. "EPDD.jl" = Fem/EllipticPdeDomainDecomposition.jl.

Conventions (differences from the reference are layout only):
  * indices are 0-based here (the reference is 1-based Julia); `-1` means "absent";
  * the reference's `Dict{Int,Int}` maps are flat integer arrays
    (`gather_idx[d][l_Γd] = l_Γ` replaces `ind_Γd_Γ2l[d]`, EPDD.jl:186-191);
  * domains are numbered 0..ndom-1; `node_owner` keeps the reference's coding shifted by
    nothing for the special values: -1 = on Γ, -2 = Dirichlet, d>=0 = interior of d
    (reference: -1 / 0 / d>=1, EPDD.jl:43-46);
  * sparse blocks are scipy CSR. The symmetric blocks (A, A_II, A_ΓΓ) have identical
    CSR and CSC arrays; `A_IΓ` (n_I x n_Γd) is kept in CSR, i.e. the CSC arrays of its
    transpose `A_ΓI`.

Deviations from the reference that are not layout (SURVEY.md N2, §2 Mesh.jl row):
  * mesh = structured triangulation of the unit square (TriangleMesh is unavailable);
  * partition = element boxes (mpmetis is unavailable);
  * interior solves in set-up (`assemble_local_schurs`, `get_schur_rhs`,
    `get_subdomain_solutions`) use a sparse direct factorisation (scipy SuperLU) where
    the reference uses `IterativeSolvers.cg` to reltol 1e-9 / sqrt(eps).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence, Union

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

Coeff = Union[np.ndarray, Callable[[np.ndarray, np.ndarray], np.ndarray]]


# --------------------------------------------------------------------------------------
# Mesh and partition (substitutes for Fem/Mesh.jl:21-31 get_mesh and :169-195 mesh_partition)
# --------------------------------------------------------------------------------------
@dataclass
class Mesh:
    cells: np.ndarray           # (3, nel) int64, 0-based, counter-clockwise
    points: np.ndarray          # (2, nnode) float64
    point_marker: np.ndarray    # (nnode,) int64, 1 = Dirichlet boundary node
    cell_neighbors: np.ndarray  # (3, nel) int64, neighbour across the edge opposite vertex j, -1 = boundary
    N: int                      # nodes per side


def get_mesh(N: int) -> Mesh:
    """Structured P1 triangulation of the unit square with N x N nodes.

    Stands in for `get_mesh(tentative_nnode)` (Fem/Mesh.jl:21-31), which calls Triangle.
    Same output fields as the TriMesh the reference consumes (`cell`, `point`,
    `point_marker`, `cell_neighbor`); every boundary node is Dirichlet as in the
    reference's `point_marker`.
    """
    if N < 3:
        raise ValueError("need N >= 3")
    xs = np.linspace(0.0, 1.0, N)
    X, Y = np.meshgrid(xs, xs, indexing="xy")          # node id = j*N + i, x = xs[i], y = xs[j]
    points = np.stack([X.ravel(), Y.ravel()]).astype(np.float64)
    ii, jj = np.meshgrid(np.arange(N), np.arange(N), indexing="xy")
    marker = ((ii == 0) | (ii == N - 1) | (jj == 0) | (jj == N - 1)).astype(np.int64).ravel()

    nc = N - 1
    ci, cj = np.meshgrid(np.arange(nc), np.arange(nc), indexing="xy")
    ci, cj = ci.ravel(), cj.ravel()                    # cell id c = cj*nc + ci
    n00 = cj * N + ci
    n10, n01, n11 = n00 + 1, n00 + N, n00 + N + 1
    nel = 2 * nc * nc
    cells = np.empty((3, nel), dtype=np.int64)
    cells[:, 0::2] = np.stack([n00, n10, n11])         # T0 of each cell
    cells[:, 1::2] = np.stack([n00, n11, n01])         # T1 of each cell
    c = cj * nc + ci
    nb = np.full((3, nel), -1, dtype=np.int64)
    # T0 = (n00, n10, n11): opposite n00 -> right cell's T1; n10 -> own T1; n11 -> lower cell's T1
    nb[0, 0::2] = np.where(ci + 1 < nc, 2 * (c + 1) + 1, -1)
    nb[1, 0::2] = 2 * c + 1
    nb[2, 0::2] = np.where(cj > 0, 2 * (c - nc) + 1, -1)
    # T1 = (n00, n11, n01): opposite n00 -> upper cell's T0; n11 -> left cell's T0; n01 -> own T0
    nb[0, 1::2] = np.where(cj + 1 < nc, 2 * (c + nc), -1)
    nb[1, 1::2] = np.where(ci > 0, 2 * (c - 1), -1)
    nb[2, 1::2] = 2 * c
    return Mesh(cells, points, marker, nb, N)


def mesh_partition(mesh: Mesh, px: int, py: int):
    """Element-box partition into px x py subdomains; returns (epart, npart), 0-based.

    Stands in for `mesh_partition(cells, ndom)` (Fem/Mesh.jl:169-195, mpmetis -contig).
    `npart[node]` is the domain of one element that owns the node, which is all
    `set_subdomains` needs (it reads npart only for nodes off the interface, EPDD.jl:168-174).
    """
    N = mesh.N
    nc = N - 1
    nel = mesh.cells.shape[1]
    c = np.arange(nel) // 2
    ci, cj = c % nc, c // nc
    bx = np.minimum(ci * px // nc, px - 1)
    by = np.minimum(cj * py // nc, py - 1)
    epart = (by * px + bx).astype(np.int64)
    npart = np.full(N * N, -1, dtype=np.int64)
    # last writer wins; any owner is valid for nodes off Γ
    for k in range(3):
        npart[mesh.cells[k]] = epart
    return epart, npart


# --------------------------------------------------------------------------------------
# Dirichlet index maps (Fem/BoundaryConditions.jl:35-57)
# --------------------------------------------------------------------------------------
@dataclass
class DirichletInds:
    dirichlet_g2l: np.ndarray       # (nnode,) -1 where not Dirichlet
    not_dirichlet_g2l: np.ndarray   # (nnode,) -1 where Dirichlet
    dirichlet_l2g: np.ndarray
    not_dirichlet_l2g: np.ndarray


def get_dirichlet_inds(points: np.ndarray, point_marker: np.ndarray) -> DirichletInds:
    """`get_dirichlet_inds` (Fem/BoundaryConditions.jl:35-57): ascending-node numbering of both sets."""
    nnode = points.shape[1]
    is_d = point_marker.ravel() == 1
    d_l2g = np.flatnonzero(is_d).astype(np.int64)
    nd_l2g = np.flatnonzero(~is_d).astype(np.int64)
    d_g2l = np.full(nnode, -1, dtype=np.int64)
    nd_g2l = np.full(nnode, -1, dtype=np.int64)
    d_g2l[d_l2g] = np.arange(d_l2g.size)
    nd_g2l[nd_l2g] = np.arange(nd_l2g.size)
    return DirichletInds(d_g2l, nd_g2l, d_l2g, nd_l2g)


def append_bc(dinds: DirichletInds, u_no_dirichlet, points, uexact):
    """`append_bc` (Fem/BoundaryConditions.jl:94-115)."""
    u = np.empty(points.shape[1])
    g = dinds.dirichlet_l2g
    u[g] = _eval(uexact, points[0, g], points[1, g])
    u[dinds.not_dirichlet_l2g] = u_no_dirichlet
    return u


def _eval(fn, x, y):
    out = fn(x, y)
    return np.broadcast_to(np.asarray(out, dtype=np.float64), np.shape(x)).astype(np.float64)


def _first_unique(a: np.ndarray) -> np.ndarray:
    """Unique values of `a` in order of first occurrence."""
    _, first = np.unique(a, return_index=True)
    return a[np.sort(first)]


# --------------------------------------------------------------------------------------
# set_subdomains (EPDD.jl:86-193)
# --------------------------------------------------------------------------------------
@dataclass
class Subdomains:
    ndom: int
    node_owner: np.ndarray              # (nnode,) -1 Γ, -2 Dirichlet, d interior
    node_Γ: np.ndarray                  # Γ-local -> global node (first-encounter order)
    node_Γ_cnt: np.ndarray              # (n_Γ,) number of subdomains sharing each Γ node
    node_Γd: List[np.ndarray]           # per domain: Γd-local -> global node (first-encounter order)
    node_Id: List[np.ndarray]           # per domain: I-local -> global node (ascending)
    gather_idx: List[np.ndarray]        # per domain: Γd-local -> Γ-local (flat ind_Γd_Γ2l)
    ind_Γ_g2l: np.ndarray               # (nnode,) global -> Γ-local or -1
    ind_I_g2l: np.ndarray               # (nnode,) global -> I-local in its owner domain or -1
    elemd: List[np.ndarray]             # elements of each domain (ascending)

    @property
    def n_Γ(self) -> int:
        return int(self.node_Γ.size)

    @property
    def n_Γd(self) -> List[int]:
        return [int(a.size) for a in self.node_Γd]

    @property
    def n_Id(self) -> List[int]:
        return [int(a.size) for a in self.node_Id]


def set_subdomains(cells, cell_neighbors, epart, npart, dirichlet_g2l) -> Subdomains:
    """`set_subdomains` (EPDD.jl:86-193).

    Γ = non-Dirichlet nodes shared by two edge-neighbouring elements of different
    subdomains (:125-152). Γ and every Γ_d are numbered by first encounter in the
    element loop (`for iel`, `for j in 1:3`, `for node in iel_cell`, :114-156);
    interiors are numbered by ascending node id within their `npart` owner (:167-181).
    """
    nel = cells.shape[1]
    nnode = int(cells.max()) + 1
    ndom = int(epart.max()) + 1
    is_dir = dirichlet_g2l >= 0

    # candidate events in reference loop order: (iel, j) with a neighbour in another domain
    jel = cell_neighbors                                    # (3, nel)
    valid = jel >= 0
    other = np.zeros_like(valid)
    other[valid] = epart[jel[valid]] != np.broadcast_to(epart, jel.shape)[valid]
    ev_j, ev_iel = np.nonzero(other)                        # row-major: sorted by j then iel
    order = np.lexsort((ev_j, ev_iel))                      # sort by iel, then j
    ev_iel, ev_j = ev_iel[order], ev_j[order]
    ev_jel = jel[ev_j, ev_iel]
    # for every event, nodes of iel (in cell order) that also belong to jel and are not Dirichlet
    icell = cells[:, ev_iel]                                # (3, nev)
    jcell = cells[:, ev_jel]
    shared = (icell[:, None, :] == jcell[None, :, :]).any(axis=1)   # (3, nev)
    shared &= ~is_dir[icell]
    k_idx, e_idx = np.nonzero(shared)
    seq = np.lexsort((k_idx, e_idx))                        # event order, then vertex order
    cand_nodes = icell[k_idx[seq], e_idx[seq]]
    cand_dom = epart[ev_iel[e_idx[seq]]]

    node_Γ = _first_unique(cand_nodes) if cand_nodes.size else np.empty(0, np.int64)
    ind_Γ_g2l = np.full(nnode, -1, dtype=np.int64)
    ind_Γ_g2l[node_Γ] = np.arange(node_Γ.size)

    node_owner = np.full(nnode, -3, dtype=np.int64)
    node_owner[is_dir] = -2
    node_owner[node_Γ] = -1

    node_Γd, gather_idx = [], []
    node_Γ_cnt = np.zeros(node_Γ.size, dtype=np.int64)
    for d in range(ndom):
        nd = cand_nodes[cand_dom == d]
        nd = _first_unique(nd) if nd.size else np.empty(0, np.int64)
        node_Γd.append(nd.astype(np.int64))
        g = ind_Γ_g2l[nd]
        gather_idx.append(g.astype(np.int64))
        node_Γ_cnt[g] += 1

    free_int = np.flatnonzero(node_owner == -3)
    node_owner[free_int] = npart[free_int]
    node_Id, ind_I_g2l = [], np.full(nnode, -1, dtype=np.int64)
    for d in range(ndom):
        nid = free_int[npart[free_int] == d]
        node_Id.append(nid.astype(np.int64))
        ind_I_g2l[nid] = np.arange(nid.size)

    elemd = [np.flatnonzero(epart == d).astype(np.int64) for d in range(ndom)]
    return Subdomains(ndom, node_owner, node_Γ.astype(np.int64), node_Γ_cnt, node_Γd, node_Id,
                      gather_idx, ind_Γ_g2l, ind_I_g2l, elemd)


# --------------------------------------------------------------------------------------
# Element kernels shared by the assemblies
# --------------------------------------------------------------------------------------
def _element_terms(cells, points, coeff: Coeff, f, uexact):
    """Per-element quantities, evaluated in the reference's operation order.

    Δa = (a1+a2+a3)/3 (EPDD.jl:260-269); shoelace terms (:272-277); Area (:280);
    ΔKij = Δa*(Δyi*Δyj + Δxi*Δxj)/4/Area (:294); Δb_i = (2f_i+f_j+f_k)*Area/12 (:340-349).
    """
    x = points[0][cells]            # (3, nel)
    y = points[1][cells]
    if callable(coeff):
        a = _eval(coeff, x, y)
    else:
        a = np.asarray(coeff, dtype=np.float64)[cells]
    Δa = np.zeros(cells.shape[1])
    for j in range(3):
        Δa = Δa + a[j]
    Δa = Δa / 3.0
    G, Area = _element_geometry(x, y)
    K = np.empty((3, 3, cells.shape[1]))
    for i in range(3):
        for j in range(3):
            K[i, j] = Δa * G[i, j] / 4 / Area
    fv = _eval(f, x, y)
    b = np.empty((3, cells.shape[1]))
    for i in range(3):
        j, k = (i + 1) % 3, (i + 2) % 3
        b[i] = (2 * fv[i] + fv[j] + fv[k]) * Area / 12
    ue = _eval(uexact, x, y)
    return K, b, ue


def _element_geometry(x, y):
    """Shoelace terms (EPDD.jl:272-277), G_ij = Δy_i*Δy_j + Δx_i*Δx_j (the coefficient-free factor of ΔK_ij) and Area (:280)."""
    Δx = np.stack([x[2] - x[1], x[0] - x[2], x[1] - x[0]])
    Δy = np.stack([y[1] - y[2], y[2] - y[0], y[0] - y[1]])
    Area = (Δx[2] * Δy[1] - Δx[1] * Δy[2]) / 2.0
    G = np.empty((3, 3, x.shape[1]))
    for i in range(3):
        for j in range(3):
            G[i, j] = Δy[i] * Δy[j] + Δx[i] * Δx[j]
    return G, Area


def _segment_sums(v, starts):
    """Sum of every segment v[starts[k]:starts[k+1]] taken strictly left to right, ((v0 + v1) + v2) + ... — the order
    in which Julia's `sparse(I,J,V)` combines duplicates and `b[k] += Δ` accumulates. (`np.add.reduceat` is NOT that:
    it adds the first term to the sum of the others.)"""
    ends = np.append(starts[1:], v.size)
    out = v[starts].copy()
    length = ends - starts
    for p in range(1, int(length.max()) if length.size else 0):
        m = length > p
        out[m] = out[m] + v[starts[m] + p]
    return out


def _coo_to_csr(I, J, V, shape, seq=None) -> sp.csr_matrix:
    """`sparse(I,J,V,m,n)`: duplicates are summed in order of occurrence (deterministic)."""
    if I.size == 0:
        return sp.csr_matrix(shape, dtype=np.float64)
    if seq is None:
        seq = np.arange(I.size)
    order = np.lexsort((seq, J, I))
    I, J, V = I[order], J[order], V[order]
    new = np.ones(I.size, dtype=bool)
    new[1:] = (I[1:] != I[:-1]) | (J[1:] != J[:-1])
    starts = np.flatnonzero(new)
    vals = _segment_sums(V, starts)
    rows, cols = I[starts], J[starts]
    indptr = np.zeros(shape[0] + 1, dtype=np.int64)
    np.add.at(indptr, rows + 1, 1)
    indptr = np.cumsum(indptr)
    m = sp.csr_matrix((vals, cols.astype(np.int64), indptr), shape=shape)
    m.has_sorted_indices = True
    return m


def _accumulate(n, idx, vals, seq):
    """b[idx] += vals applied in `seq` order (sequential accumulation per entry)."""
    out = np.zeros(n)
    if idx.size:
        order = np.lexsort((seq, idx))
        i, v = idx[order], vals[order]
        new = np.ones(i.size, dtype=bool)
        new[1:] = i[1:] != i[:-1]
        starts = np.flatnonzero(new)
        out[i[starts]] = _segment_sums(v, starts)
    return out


# --------------------------------------------------------------------------------------
# Full-system assembly (Fem/EllipticPde.jl:60-157 function coeff, :187-275 nodal coeff)
# --------------------------------------------------------------------------------------
def do_isotropic_elliptic_assembly(cells, points, dinds: DirichletInds, point_marker,
                                   a: Coeff, f, uexact):
    """Full Galerkin system on the non-Dirichlet nodes: returns (A csr, b)."""
    K, be, ue = _element_terms(cells, points, a, f, uexact)
    nel = cells.shape[1]
    n = dinds.not_dirichlet_l2g.size
    is_dir = point_marker.ravel()[cells] == 1       # (3, nel)
    loc = dinds.not_dirichlet_g2l[cells]            # -1 on Dirichlet
    I, J, V, S = [], [], [], []
    bi, bv, bs = [], [], []
    el = np.arange(nel)
    for i in range(3):
        for j in range(3):
            both = ~is_dir[i] & ~is_dir[j]
            I.append(loc[i][both]); J.append(loc[j][both]); V.append(K[i, j][both])
            S.append(el[both] * 9 + i * 3 + j)
            lift = is_dir[i] & ~is_dir[j]
            bi.append(loc[j][lift]); bv.append(-(ue[i][lift] * K[i, j][lift]))
            bs.append(el[lift] * 12 + i * 3 + j)
    for i in range(3):
        free = ~is_dir[i]
        bi.append(loc[i][free]); bv.append(be[i][free]); bs.append(el[free] * 12 + 9 + i)
    A = _coo_to_csr(np.concatenate(I), np.concatenate(J), np.concatenate(V), (n, n), np.concatenate(S))
    b = _accumulate(n, np.concatenate(bi), np.concatenate(bv), np.concatenate(bs))
    return A, b


# --------------------------------------------------------------------------------------
# prepare_local_schurs / prepare_global_schur (EPDD.jl:389-546 / 212-369)
# --------------------------------------------------------------------------------------
def _triplets(cells, sub: Subdomains, local: bool):
    """Index side of the element loop (EPDD.jl:283-350 / 459-527): for every block the COO (row, col) pairs and, for
    every pair, its sequence code s = 12*element + 3*i + j (stiffness term ΔK_ij, also the Dirichlet lifting when it
    lands in a right-hand side) or 12*element + 9 + i (load term Δb_i). The code is both the reference's order of
    accumulation and the address of the value, so the numeric side (host: `_values`, device: `AssemblyPlan`) needs
    nothing else."""
    nel = cells.shape[1]
    owner = sub.node_owner[cells]                   # (3, nel)
    el = np.arange(nel)
    gΓ = sub.ind_Γ_g2l[cells]
    gI = sub.ind_I_g2l[cells]
    # Γ_d-local index of every (vertex, element): only meaningful where owner == -1
    gΓd = np.full(cells.shape, -1, dtype=np.int64)
    tmp = np.full(sub.node_owner.size, -1, dtype=np.int64)
    for d in range(sub.ndom):
        tmp[sub.node_Γd[d]] = np.arange(sub.node_Γd[d].size)
        e = sub.elemd[d]
        gΓd[:, e] = tmp[cells[:, e]]
        tmp[sub.node_Γd[d]] = -1
    cat = np.concatenate
    out = dict(II=[], IΓ=[], ΓΓ=[], bI=[], bΓ=None, ΓΓ_glob=None)
    bΓ_i, bΓ_s = [], []
    ΓΓ_glob = ([], [], [])
    for d in range(sub.ndom):
        e = sub.elemd[d]
        ow = owner[:, e]
        gId, gΓe, gΓde, ele = gI[:, e], gΓ[:, e], gΓd[:, e], el[e]
        II = ([], [], []); IΓ = ([], [], []); ΓΓ = ([], [], [])
        bI = ([], [])
        for i in range(3):
            for j in range(3):
                s = ele * 12 + i * 3 + j
                oi, oj = ow[i], ow[j]
                m = (oi == -1) & (oj == -1)
                if local:
                    ΓΓ[0].append(gΓde[i][m]); ΓΓ[1].append(gΓde[j][m]); ΓΓ[2].append(s[m])
                else:
                    ΓΓ_glob[0].append(gΓe[i][m]); ΓΓ_glob[1].append(gΓe[j][m]); ΓΓ_glob[2].append(s[m])
                m = (oi >= 0) & (oj >= 0)
                II[0].append(gId[i][m]); II[1].append(gId[j][m]); II[2].append(s[m])
                m = (oi >= 0) & (oj == -1)
                IΓ[0].append(gId[i][m]); IΓ[1].append((gΓde if local else gΓe)[j][m]); IΓ[2].append(s[m])
                # Dirichlet lifting (EPDD.jl:322-331 / 498-507): b[j] -= ΔK_ij * uexact(node i)
                m = (oi == -2) & (oj == -1)
                bΓ_i.append(gΓe[j][m]); bΓ_s.append(s[m])
                m = (oi == -2) & (oj >= 0)
                bI[0].append(gId[j][m]); bI[1].append(s[m])
        for i in range(3):
            s = ele * 12 + 9 + i
            m = ow[i] == -1
            bΓ_i.append(gΓe[i][m]); bΓ_s.append(s[m])
            m = ow[i] >= 0
            bI[0].append(gId[i][m]); bI[1].append(s[m])
        out["II"].append(tuple(cat(v) for v in II))
        out["IΓ"].append(tuple(cat(v) for v in IΓ))
        if local:
            out["ΓΓ"].append(tuple(cat(v) for v in ΓΓ))
        out["bI"].append(tuple(cat(v) for v in bI))
    out["bΓ"] = (cat(bΓ_i), cat(bΓ_s))
    if not local:
        out["ΓΓ_glob"] = tuple(cat(v) for v in ΓΓ_glob)
    return out


def _values(seq, K, be, ue, rhs: bool):
    """Value of every contribution from its sequence code: ΔK_ij; in a right-hand side -(ΔK_ij * uexact_i) or Δb_i."""
    e, c = seq // 12, seq % 12
    if not rhs:
        return K.reshape(9, -1)[c, e]
    lift = c < 9
    cl = np.where(lift, c, 0)
    v = -(K.reshape(9, -1)[cl, e] * ue[cl // 3, e])
    return np.where(lift, v, be[np.where(lift, 0, c - 9), e])


def _prepare(cells, points, epart, sub: Subdomains, coeff, f, uexact, local: bool):
    K, be, ue = _element_terms(cells, points, coeff, f, uexact)
    T = _triplets(cells, sub, local)
    n_Γ = sub.n_Γ
    A_II, A_IΓ, A_ΓΓ_parts, b_Id = [], [], [], []
    for d in range(sub.ndom):
        nI, nΓd = sub.node_Id[d].size, sub.node_Γd[d].size
        I, J, S = T["II"][d]
        A_II.append(_coo_to_csr(I, J, _values(S, K, be, ue, False), (nI, nI), S))
        I, J, S = T["IΓ"][d]
        A_IΓ.append(_coo_to_csr(I, J, _values(S, K, be, ue, False), (nI, nΓd if local else n_Γ), S))
        if local:
            I, J, S = T["ΓΓ"][d]
            A_ΓΓ_parts.append(_coo_to_csr(I, J, _values(S, K, be, ue, False), (nΓd, nΓd), S))
        I, S = T["bI"][d]
        b_Id.append(_accumulate(nI, I, _values(S, K, be, ue, True), S))
    I, S = T["bΓ"]
    b_Γ = _accumulate(n_Γ, I, _values(S, K, be, ue, True), S)
    if local:
        return A_II, A_IΓ, A_ΓΓ_parts, b_Id, b_Γ
    I, J, S = T["ΓΓ_glob"]
    A_ΓΓ = _coo_to_csr(I, J, _values(S, K, be, ue, False), (n_Γ, n_Γ), S)
    return A_II, A_IΓ, A_ΓΓ, b_Id, b_Γ


def prepare_local_schurs(cells, points, epart, sub: Subdomains, coeff: Coeff, f, uexact):
    """`prepare_local_schurs` (EPDD.jl:389-546): (A_IIdd, A_IΓdd, A_ΓΓdd, b_Id, b_Γ), Γ_d-local columns."""
    return _prepare(cells, points, epart, sub, coeff, f, uexact, local=True)


def prepare_global_schur(cells, points, epart, sub: Subdomains, coeff: Coeff, f, uexact):
    """`prepare_global_schur` (EPDD.jl:212-369): (A_IId, A_IΓd, A_ΓΓ, b_Id, b_Γ), Γ-global columns."""
    return _prepare(cells, points, epart, sub, coeff, f, uexact, local=False)


@dataclass
class AssemblyPlan:
    """The index half of `prepare_local_schurs` (EPDD.jl:389-546) for a FIXED mesh, partition, `f` and `uexact`: what
    Example07's realization loop (:162-171) recomputes for every coefficient draw although only `a` changes.

    Output entries, in this order: the stored values of A_IIdd[0..ndom), A_IΓdd[0..ndom), A_ΓΓdd[0..ndom) — the `nzval`
    of Julia's CSC matrices, which is also what `mi_schur_matfree_*create` / `_set_values` take — then b_Id[0..ndom), b_Γ. Entry k is the sum, in the reference's
    element order, of the contributions `ccode[cptr[k]:cptr[k+1]]`; a contribution code is 12*element + 3*i + j
    (ΔK_ij = Δa*G_ij/4/Area; in a right-hand side: -(ΔK_ij*ue_i)) or 12*element + 9 + i (Δb_i, coefficient-free).
    `api.AssemblyPlan(ctx, plan)` runs it on the GPU; `blocks(values)` rebuilds the scipy matrices on the host."""
    cells: np.ndarray
    G: np.ndarray          # (9, nel)
    area: np.ndarray       # (nel,)
    ue: np.ndarray         # (3, nel)
    be: np.ndarray         # (3, nel)
    cptr: np.ndarray
    ccode: np.ndarray
    n_matrix_entries: int
    layout: dict           # name -> list of (offset, count) per subdomain (b_Γ: single tuple)
    patterns: dict         # name -> list of (indptr, indices, shape)
    n_node: int

    @property
    def n_entries(self) -> int:
        return self.cptr.size - 1

    def blocks(self, values: np.ndarray):
        """(A_IIdd, A_IΓdd, A_ΓΓdd, b_Id, b_Γ) from a flat value array, as `prepare_local_schurs` returns them."""
        values = np.asarray(values, dtype=np.float64)
        out = []
        for name in ("II", "IΓ", "ΓΓ"):
            mats = []
            for (off, cnt), (indptr, indices, shape) in zip(self.layout[name], self.patterns[name]):
                if name == "IΓ":
                    m = sp.csc_matrix((values[off:off + cnt].copy(), indices, indptr), shape=shape).tocsr()
                else:
                    m = sp.csr_matrix((values[off:off + cnt].copy(), indices, indptr), shape=shape)
                m.has_sorted_indices = True
                mats.append(m)
            out.append(mats)
        out.append([values[off:off + cnt].copy() for off, cnt in self.layout["bI"]])
        off, cnt = self.layout["bΓ"]
        out.append(values[off:off + cnt].copy())
        return tuple(out)


def make_assembly_plan(cells, points, epart, sub: Subdomains, f, uexact) -> AssemblyPlan:
    x, y = points[0][cells], points[1][cells]
    G, Area = _element_geometry(x, y)
    fv = _eval(f, x, y)
    be = np.empty((3, cells.shape[1]))
    for i in range(3):
        j, k = (i + 1) % 3, (i + 2) % 3
        be[i] = (2 * fv[i] + fv[j] + fv[k]) * Area / 12
    ue = _eval(uexact, x, y)
    T = _triplets(cells, sub, local=True)
    counts, codes = [], []
    layout = dict(II=[], IΓ=[], ΓΓ=[], bI=[], bΓ=None)
    patterns = dict(II=[], IΓ=[], ΓΓ=[])
    off = 0
    for name in ("II", "IΓ", "ΓΓ"):
        for d in range(sub.ndom):
            I, J, S = T[name][d]
            nI, nΓd = sub.node_Id[d].size, sub.node_Γd[d].size
            shape = dict(II=(nI, nI), IΓ=(nI, nΓd), ΓΓ=(nΓd, nΓd))[name]
            if name == "IΓ":
                I, J = J, I            # A_IΓdd is stored by COLUMN (the CSC arrays the C ABI takes = CSR of A_ΓIdd)
            order = np.lexsort((S, J, I))
            I, J, S = I[order], J[order], S[order]
            new = np.ones(I.size, dtype=bool)
            new[1:] = (I[1:] != I[:-1]) | (J[1:] != J[:-1])
            starts = np.flatnonzero(new)
            indptr = np.zeros((shape[1] if name == "IΓ" else shape[0]) + 1, dtype=np.int64)
            np.add.at(indptr, I[starts] + 1, 1)
            patterns[name].append((np.cumsum(indptr), J[starts].astype(np.int64), shape))
            counts.append(np.diff(np.append(starts, I.size)))
            codes.append(S)
            layout[name].append((off, starts.size))
            off += starts.size
    n_matrix = off
    rhs = [(T["bI"][d], sub.node_Id[d].size) for d in range(sub.ndom)] + [(T["bΓ"], sub.n_Γ)]
    for k, ((idx, S), n) in enumerate(rhs):
        order = np.lexsort((S, idx))
        counts.append(np.bincount(idx, minlength=n).astype(np.int64))
        codes.append(S[order])
        if k < sub.ndom:
            layout["bI"].append((off, n))
        else:
            layout["bΓ"] = (off, n)
        off += n
    cptr = np.zeros(off + 1, dtype=np.int64)
    np.cumsum(np.concatenate(counts), out=cptr[1:])
    return AssemblyPlan(np.ascontiguousarray(cells, dtype=np.int64), np.ascontiguousarray(G.reshape(9, -1)),
                        np.ascontiguousarray(Area), np.ascontiguousarray(ue), np.ascontiguousarray(be), cptr,
                        np.concatenate(codes).astype(np.int64), n_matrix, layout, patterns, int(points.shape[1]))


# --------------------------------------------------------------------------------------
# Interior solves and assembled local Schur complements
# --------------------------------------------------------------------------------------
class InteriorSolver:
    """Host-side A_II^{-1} (sparse direct). Stands in for the reference's
    `IterativeSolvers.cg(A_II, rhs; Pl=AMG, reltol)` (EPDD.jl:648-650) and is what
    BASELINE.json's north_star calls "the host-side sparse Cholesky"."""

    def __init__(self, A_II: sp.spmatrix):
        self.n = A_II.shape[0]
        self.lu = spla.splu(sp.csc_matrix(A_II), permc_spec="MMD_AT_PLUS_A",
                            diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))

    def __call__(self, rhs: np.ndarray) -> np.ndarray:
        return self.lu.solve(np.ascontiguousarray(rhs, dtype=np.float64))


class LazySolvers:
    """`solvers[d]`: the InteriorSolver of subdomain d, factorised on first use (d outside [lo, hi) -> None)."""

    def __init__(self, A_II, lo, hi):
        self._A, self._lo, self._hi, self._f = A_II, lo, hi, {}

    def __len__(self):
        return len(self._A)

    def __getitem__(self, d):
        if isinstance(d, slice):
            return [self[k] for k in range(*d.indices(len(self)))]
        if d < 0:
            d += len(self._A)
        if not (self._lo <= d < self._hi):
            return None
        if d not in self._f:
            self._f[d] = InteriorSolver(self._A[d])
        return self._f[d]

    def __iter__(self):
        return (self[d] for d in range(len(self._A)))


def _bfs_levels_from_interface(A_II: sp.csr_matrix, seeds: np.ndarray):
    """Breadth-first levels of the graph of A_II starting from `seeds`; nodes that are not
    connected to the seeds are left out (they cannot influence the Schur complement)."""
    n = A_II.shape[0]
    level = np.full(n, -1, dtype=np.int64)
    level[seeds] = 0
    frontier, levels = np.asarray(seeds, dtype=np.int64), []
    indptr, indices = A_II.indptr, A_II.indices
    while frontier.size:
        levels.append(frontier)
        starts, ends = indptr[frontier], indptr[frontier + 1]
        cnt = ends - starts
        idx = np.repeat(starts - np.concatenate(([0], np.cumsum(cnt)[:-1])), cnt) + np.arange(cnt.sum())
        nb = np.unique(indices[idx])
        nb = nb[level[nb] < 0]
        level[nb] = len(levels)
        frontier = nb
    return levels


def _dense_device():
    """Where the dense level recursion runs: the GPU through torch when one is visible (set-up
    only — rocBLAS/rocSOLVER library calls, not part of the hot path), else torch on the CPU,
    else numpy. scipy.linalg is avoided: its bundled OpenBLAS stalls for ~0.5 s per small
    Cholesky when 8+ threads spin inside a container."""
    try:
        import torch
    except ImportError:
        return None, None
    import os
    if torch.cuda.is_available() and not os.environ.get("MI355_SETUP_ON_CPU"):
        return torch, torch.device("cuda")
    return torch, torch.device("cpu")


def local_schur_by_level_elimination(A_II, A_IΓ, A_ΓΓ, b_I=None):
    """Dense S_d = A_ΓΓ - A_IΓ' A_II^{-1} A_IΓ by exact block elimination; with `b_I` also the condensed
    right-hand side A_IΓ' A_II^{-1} b_I (what `get_schur_rhs` subtracts from b_Γ, EPDD.jl:853-861), returned as
    (S_d, w_d).

    The interior is split into breadth-first levels L_0, L_1, ... grown from the interior nodes
    adjacent to Γ_d; A_II is block tridiagonal in that ordering, so eliminating from the deepest
    level towards Γ_d is the recursion T_m = A_mm, T_k = A_kk - A_{k+1,k}' T_{k+1}^{-1} A_{k+1,k}
    (and g_m = b_m, g_k = b_k - A_{k+1,k}' T_{k+1}^{-1} g_{k+1}), and S_d = A_ΓΓ - A_{0Γ}' T_0^{-1} A_{0Γ},
    w_d = A_{0Γ}' T_0^{-1} g_0. Every step is dense Cholesky + triangular solve + product (BLAS-3), which is
    far faster than n_Γd sparse triangular solves with 124 k unknowns. Interior nodes not connected to Γ_d
    cannot influence either result and are left out.
    """
    A_II = sp.csr_matrix(A_II)
    A_IΓ = sp.csr_matrix(A_IΓ)
    S = np.asarray(sp.csr_matrix(A_ΓΓ).todense(), dtype=np.float64)
    seeds = np.flatnonzero(np.diff(A_IΓ.indptr) > 0)
    if seeds.size == 0:
        return S if b_I is None else (S, np.zeros(S.shape[0]))
    levels = _bfs_levels_from_interface(A_II, seeds)
    perm = np.concatenate(levels)
    off = np.concatenate(([0], np.cumsum([l.size for l in levels])))
    Ap = sp.csr_matrix(A_II[perm][:, perm])
    torch, dev = _dense_device()

    def to_dev(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev) if torch is not None else a

    def blk(i, j):
        return to_dev(Ap[off[i]:off[i + 1], off[j]:off[j + 1]].toarray())

    if torch is not None:
        chol = torch.linalg.cholesky
        def trsm(c, b):
            return torch.linalg.solve_triangular(c, b, upper=False)
    else:
        import scipy.linalg as sla
        chol = np.linalg.cholesky
        def trsm(c, b):
            return sla.solve_triangular(c, b, lower=True, check_finite=False)

    bp = None if b_I is None else np.asarray(b_I, dtype=np.float64)[perm]
    m = len(levels) - 1
    T = blk(m, m)
    g = None if bp is None else to_dev(bp[off[m]:off[m + 1]].reshape(-1, 1))
    for k in range(m - 1, -1, -1):
        c = chol(T)
        C = blk(k + 1, k)
        Y = trsm(c, C if g is None else (torch.cat([C, g], 1) if torch is not None else np.hstack([C, g])))
        Yc = Y if g is None else Y[:, :-1]
        T = blk(k, k) - Yc.T @ Yc
        if g is not None:
            g = to_dev(bp[off[k]:off[k + 1]].reshape(-1, 1)) - Yc.T @ Y[:, -1:]
    c = chol(T)
    B = to_dev(A_IΓ[levels[0]].toarray())
    Y = trsm(c, B if g is None else (torch.cat([B, g], 1) if torch is not None else np.hstack([B, g])))
    Yc = Y if g is None else Y[:, :-1]
    YtY = Yc.T @ Yc
    S -= YtY.cpu().numpy() if torch is not None else YtY
    if g is None:
        return S
    w = Yc.T @ Y[:, -1:]
    return S, (w.cpu().numpy() if torch is not None else w).ravel()


def assemble_local_schurs(A_IIdd, A_IΓdd, A_ΓΓdd, solvers: Optional[Sequence["InteriorSolver"]] = None,
                          method: str = "levels", chunk: int = 256, b_Id=None):
    """`assemble_local_schurs` (EPDD.jl:667-695): dense S_d = A_ΓΓd - A_IΓd' A_IId^{-1} A_IΓd.

    The reference applies `apply_local_schur` (interior CG, reltol 1e-9) to every unit vector and
    keeps the upper triangle (`Symmetric(Array(map))`, :692). Here the blocks come from an exact
    direct elimination (`method="levels"`, see local_schur_by_level_elimination) or from multi-RHS
    SuperLU solves (`method="solves"`, slow; kept as the cross-check), and are symmetrised the
    same way. Returns column-major (Fortran-order) arrays like Julia's `Array`; with `b_Id` (levels only)
    also the list of condensed right-hand sides A_IΓd' A_IId^{-1} b_Id.
    """
    out, ws = [], []
    for d in range(len(A_IIdd)):
        n = A_ΓΓdd[d].shape[0]
        if method == "levels":
            res = local_schur_by_level_elimination(A_IIdd[d], A_IΓdd[d], A_ΓΓdd[d], None if b_Id is None else b_Id[d])
            S = res if b_Id is None else res[0]
            if b_Id is not None:
                ws.append(res[1])
        else:
            solve = solvers[d] if solvers is not None else InteriorSolver(A_IIdd[d])
            S = np.asarray(A_ΓΓdd[d].todense(), dtype=np.float64)
            AIΓ = sp.csc_matrix(A_IΓdd[d])
            AΓI = sp.csr_matrix(A_IΓdd[d].T)
            for c0 in range(0, n, chunk):
                c1 = min(n, c0 + chunk)
                V = solve(AIΓ[:, c0:c1].toarray())
                S[:, c0:c1] -= AΓI @ V
        S = np.triu(S) + np.triu(S, 1).T
        out.append(np.asfortranarray(S))
    return out if b_Id is None else (out, ws)


def prepare_neumann_neumann_schur_precond(Sd: Sequence[np.ndarray]):
    """`prepare_neumann_neumann_schur_precond(Sd_local_mat, ...)` (EPDD.jl:1201-1220):
    ΠS_d = pinv(S_d, rtol = sqrt(eps(Float64))) — SVD-based, singular values <= rtol*σ_max dropped.
    Runs on the GPU through torch when one is visible (set-up only), else numpy; same definition."""
    rtol = float(np.sqrt(np.finfo(np.float64).eps))
    torch, dev = _dense_device()
    out = []
    for S in Sd:
        S = np.asarray(S, dtype=np.float64)
        if torch is not None and dev.type == "cuda" and S.shape[0] > 0:
            P = torch.linalg.pinv(torch.from_numpy(np.ascontiguousarray(S)).to(dev), rtol=rtol, hermitian=False).cpu().numpy()
        else:
            P = np.linalg.pinv(S, rcond=rtol)
        out.append(np.asfortranarray(P))
    return out


def get_schur_rhs(b_Id, A_IId, A_IΓd, b_Γ, gather_idx=None, solvers=None):
    """`get_schur_rhs` (EPDD.jl:798-821 global columns; :835-864 local columns + scatter)."""
    b = np.array(b_Γ, dtype=np.float64, copy=True)
    for d in range(len(b_Id)):
        solve = solvers[d] if solvers is not None else InteriorSolver(A_IId[d])
        w = A_IΓd[d].T @ solve(b_Id[d])
        if gather_idx is None:
            b -= w
        else:
            b[gather_idx[d]] -= w
    return b


def get_subdomain_solutions(u_Γ, A_IId, A_IΓd_global, b_Id, solvers=None):
    """`get_subdomain_solutions` (EPDD.jl:1014-1025): u_I = A_II^{-1}(b_I - A_IΓ u_Γ)."""
    out = []
    for d in range(len(b_Id)):
        solve = solvers[d] if solvers is not None else InteriorSolver(A_IId[d])
        out.append(solve(b_Id[d] - A_IΓd_global[d] @ u_Γ))
    return out


def merge_subdomain_solutions(u_Γ, u_Id, sub: Subdomains, dinds: DirichletInds, uexact, points):
    """`merge_subdomain_solutions` (EPDD.jl:1040-1070)."""
    u = np.empty(points.shape[1])
    u[sub.node_Γ] = u_Γ
    for d in range(sub.ndom):
        u[sub.node_Id[d]] = u_Id[d]
    g = dinds.dirichlet_l2g
    u[g] = _eval(uexact, points[0, g], points[1, g])
    return u


# --------------------------------------------------------------------------------------
# Synthetic lognormal coefficient (semantics of
# `draw!` Fem/KarhunenLoeveDomainDecomposition.jl:1026-1044: g = Σ sqrt(Λ_α) ξ_α Ψ[:,α])
# --------------------------------------------------------------------------------------
@dataclass
class SyntheticKL:
    Λ: np.ndarray       # (m,)
    Ψ: np.ndarray       # (nnode, m)


def synthetic_kl(points, m_side: int = 8, L: float = 0.1, sig2: float = 1.0) -> SyntheticKL:
    """Separable cosine modes Ψ_{kl} = cos(kπx)cos(lπy), Λ_{kl} ∝ exp(-(k²+l²)π²L²/4),
    scaled so that Σ Λ = sig2; m = m_side² modes ordered by decreasing Λ (ties: k then l)."""
    ks, ls = np.meshgrid(np.arange(m_side), np.arange(m_side), indexing="ij")
    ks, ls = ks.ravel(), ls.ravel()
    lam = np.exp(-(ks ** 2 + ls ** 2) * np.pi ** 2 * L ** 2 / 4)
    order = np.lexsort((ls, ks, -lam))
    ks, ls, lam = ks[order], ls[order], lam[order]
    lam = lam * (sig2 / lam.sum())
    x, y = points
    Ψ = np.cos(np.pi * x[:, None] * ks[None, :]) * np.cos(np.pi * y[:, None] * ls[None, :])
    return SyntheticKL(lam, Ψ)


def draw(kl: SyntheticKL, rng: np.random.Generator):
    """`draw!`: ξ ~ N(0, I); g = Σ_α sqrt(Λ_α) ξ_α Ψ[:,α] accumulated mode by mode."""
    ξ = rng.standard_normal(kl.Λ.size)
    g = np.zeros(kl.Ψ.shape[0])
    for α in range(kl.Λ.size):
        g += (np.sqrt(kl.Λ[α]) * ξ[α]) * kl.Ψ[:, α]
    return ξ, g


# --------------------------------------------------------------------------------------
# One-call problem builders used by tests, smoke() and bench.py
# --------------------------------------------------------------------------------------
@dataclass
class SchurProblem:
    mesh: Mesh
    dinds: DirichletInds
    sub: Subdomains
    epart: np.ndarray
    A_IIdd: list
    A_IΓdd: list
    A_ΓΓdd: list
    b_Id: list
    b_Γ: np.ndarray
    b_schur: np.ndarray
    Sd: Optional[list] = None       # dense local Schur complements, column-major
    ΠSd: Optional[list] = None      # their pseudo-inverses, column-major
    solvers: Optional[list] = None
    uexact: Optional[Callable] = None
    info: dict = field(default_factory=dict)


def build_schur_problem(N: int, px: int, py: int, coeff: Coeff, f, uexact,
                        assemble: bool = True, precond: bool = True, dom_slice=None,
                        mesh: Optional[Mesh] = None, partition=None, blocks=None, sub: Optional[Subdomains] = None,
                        dense_setup=None) -> SchurProblem:
    """Example03:45-150 set-up flow on the synthetic mesh (mesh → partition → maps →
    local blocks → b_schur → assembled S_d → Neumann-Neumann pseudo-inverses).

    `dom_slice=(lo, hi)` (multi-GPU: one call per rank) factorises, assembles and pseudo-inverts
    only subdomains lo..hi-1; the other entries of `solvers`, `Sd`, `ΠSd` are None and
    `b_schur` then holds only this rank's share  -Σ_{d in slice} R_d' A_IΓd' A_IId^{-1} b_Id
    (+ b_Γ on the rank that owns subdomain 0), so that the sum over ranks is the full b_schur.
    `mesh` / `partition=(epart, npart)` replace the structured substitutes, e.g. with files written by the
    reference's Triangle + METIS pipeline (io.load_mesh / io.load_partition).
    `blocks=(A_IIdd, A_IΓdd, A_ΓΓdd, b_Id, b_Γ)` skips the element loop (e.g. `AssemblyPlan.blocks` of values
    assembled on the GPU; `coeff` is then unused) and `sub` re-uses the subdomain maps of an earlier call.
    """
    mesh = get_mesh(N) if mesh is None else mesh
    dinds = get_dirichlet_inds(mesh.points, mesh.point_marker)
    epart, npart = mesh_partition(mesh, px, py) if partition is None else partition
    if sub is None:
        sub = set_subdomains(mesh.cells, mesh.cell_neighbors, epart, npart, dinds.dirichlet_g2l)
    if blocks is None:
        blocks = prepare_local_schurs(mesh.cells, mesh.points, epart, sub, coeff, f, uexact)
    A_II, A_IΓ, A_ΓΓ, b_Id, b_Γ = blocks
    lo, hi = (0, sub.ndom) if dom_slice is None else dom_slice
    loc = range(lo, hi)
    b_schur = np.array(b_Γ, copy=True) if lo == 0 else np.zeros_like(b_Γ)
    solvers = LazySolvers(A_II, lo, hi)          # SuperLU factors, built only if something asks for them
    Sd = Pi = None
    if assemble:
        # one elimination per subdomain gives S_d and the condensed rhs (get_schur_rhs, EPDD.jl:835-864)
        # `dense_setup(A_II, A_IΓ, A_ΓΓ, b_Id, want_pinv) -> (S_d list, w_d list, ΠS_d list | None)`: e.g. the device set-up of the
        # library (api.device_dense_setup: mi_schur_setup_run + mi_nn_pinv) in place of this module's host elimination
        Pl = None
        if dense_setup is not None:
            Sl, wl, Pl = dense_setup([A_II[d] for d in loc], [A_IΓ[d] for d in loc], [A_ΓΓ[d] for d in loc],
                                     [b_Id[d] for d in loc], precond)
        else:
            Sl, wl = assemble_local_schurs([A_II[d] for d in loc], [A_IΓ[d] for d in loc], [A_ΓΓ[d] for d in loc],
                                           b_Id=[b_Id[d] for d in loc])
        for d in loc:
            b_schur[sub.gather_idx[d]] -= wl[d - lo]
        Sd = [Sl[d - lo] if lo <= d < hi else None for d in range(sub.ndom)]
        if precond:
            if Pl is None:
                Pl = prepare_neumann_neumann_schur_precond(Sl)
            Pi = [Pl[d - lo] if lo <= d < hi else None for d in range(sub.ndom)]
    else:
        for d in loc:                           # get_schur_rhs with sparse direct interior solves
            b_schur[sub.gather_idx[d]] -= A_IΓ[d].T @ solvers[d](b_Id[d])
    prob = SchurProblem(mesh, dinds, sub, epart, A_II, A_IΓ, A_ΓΓ, b_Id, b_Γ, b_schur,
                        solvers=solvers, uexact=uexact)
    prob.info["dom_slice"] = (lo, hi)
    prob.Sd, prob.ΠSd = Sd, Pi
    return prob

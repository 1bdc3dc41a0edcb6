"""The reference's on-disk formats, so that meshes / partitions produced by the Julia
pipeline (Triangle + METIS) can be fed to this library and its outputs read by the reference's `*.py` plots.

NPZ.jl's `npzwrite(filename, array)` writes ONE array in NPY format whatever the file extension is, so the
reference's `data/*.npz` files are NPY files; numpy.load recognises them by their magic bytes.

  Fem/Mesh.jl:49-55   save_mesh:       cells' .- 1 (nel x 3, 0-based), points' (nnode x 2),
                                       point_markers' (nnode x 1), cell_neighbors' (nel x 3, as TriangleMesh gives them)
  Fem/Mesh.jl:84-91   load_mesh:       the inverse (cells .+ 1)
  Fem/Mesh.jl:216-219 save_partition:  epart .- 1, npart .- 1
  Example07:281-285   iteration counts as a 1-D integer array

In memory this package is 0-based with -1 as the boundary tag of `cell_neighbors` (fem.Mesh).
"""
from __future__ import annotations

import os

import numpy as np

from .fem import Mesh


def _write(path: str, a: np.ndarray) -> None:
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as fh:           # a file object: numpy must not append ".npy"
        np.save(fh, np.asfortranarray(a))  # Julia arrays are column-major; NPZ.jl sets fortran_order


def mesh_paths(data_dir: str, tentative_nnode: int):
    root = os.path.join(data_dir, f"DoF{tentative_nnode}")
    return {k: f"{root}.{k}.npz" for k in ("cells", "points", "point_markers", "cell_neighbors")}


def save_mesh(mesh: Mesh, tentative_nnode: int, data_dir: str = "data") -> None:
    """`save_mesh(mesh, tentative_nnode)` (Fem/Mesh.jl:49-55). Neighbours are written 1-based with -1 on the
    boundary, the convention `set_subdomains` expects after its TriangleMesh correction (EPDD.jl:106-111)."""
    p = mesh_paths(data_dir, tentative_nnode)
    _write(p["cells"], mesh.cells.T.astype(np.int64))
    _write(p["points"], mesh.points.T.astype(np.float64))
    _write(p["point_markers"], mesh.point_marker.reshape(-1, 1).astype(np.int64))
    nb = np.where(mesh.cell_neighbors < 0, -1, mesh.cell_neighbors + 1)
    _write(p["cell_neighbors"], nb.T.astype(np.int64))


def load_mesh(tentative_nnode: int, data_dir: str = "data") -> Mesh:
    """`load_mesh(tentative_nnode)` (Fem/Mesh.jl:84-91) + the neighbour-index correction of
    `set_subdomains` (EPDD.jl:106-111: if the largest neighbour index exceeds nel the table is shifted by one
    and the boundary tag is -1; otherwise the smallest entry is the boundary tag)."""
    p = mesh_paths(data_dir, tentative_nnode)
    cells = np.load(p["cells"]).astype(np.int64).T.copy()                 # (3, nel), already 0-based on disk
    points = np.load(p["points"]).astype(np.float64).T.copy()             # (2, nnode)
    marker = np.load(p["point_markers"]).astype(np.int64).ravel()
    nb = np.load(p["cell_neighbors"]).astype(np.int64).T.copy()           # (3, nel), Julia/Triangle numbering
    nel = cells.shape[1]
    bnd_tag, iel_max = int(nb.min()), int(nb.max())
    if iel_max > nel:
        nb = nb - 1
        bnd_tag = -1
    nb0 = np.where(nb == bnd_tag, -1, nb - 1)                             # 1-based -> 0-based, -1 = boundary
    n_side = int(round(np.sqrt(points.shape[1])))
    return Mesh(cells, points, marker, nb0, n_side)


def partition_paths(data_dir: str, tentative_nnode: int, ndom: int):
    root = os.path.join(data_dir, f"DoF{tentative_nnode}-ndom{ndom}")
    return f"{root}.epart.npz", f"{root}.npart.npz"


def save_partition(epart, npart, tentative_nnode: int, ndom: int, data_dir: str = "data") -> None:
    """`save_partition` (Fem/Mesh.jl:216-219): 0-based subdomain ids on disk."""
    pe, pn = partition_paths(data_dir, tentative_nnode, ndom)
    _write(pe, np.asarray(epart, dtype=np.int64))
    _write(pn, np.asarray(npart, dtype=np.int64))


def load_partition(tentative_nnode: int, ndom: int, data_dir: str = "data"):
    """`load_partition` (Fem/Mesh.jl:243-247); returned 0-based (the reference adds 1 for Julia)."""
    pe, pn = partition_paths(data_dir, tentative_nnode, ndom)
    return np.load(pe).astype(np.int64).ravel(), np.load(pn).astype(np.int64).ravel()


def save_pcg_iters(iters, root_fname: str, ndom: int, tag: str, nreals: int, data_dir: str = "data") -> str:
    """Example07:284-285: `data/$root_fname.neumann-neumann_ndom$ndom_$tag.pcg-iters.nreals$nreals.npz`."""
    path = os.path.join(data_dir, f"{root_fname}.neumann-neumann_ndom{ndom}_{tag}.pcg-iters.nreals{nreals}.npz")
    _write(path, np.asarray(iters, dtype=np.int64))
    return path

import os
import sys

# In-process ranks (tests/test_gpu_multirank.py, test_gpu_parity.py) join their contexts by device-side flags: kernels of
# one rank wait for kernels of another, which needs a hardware queue per rank (include/mi355schur.h, mi_ctx_loopback_init).
# The HIP runtime reads this once, at its first call.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
# ... and the ranks' host<->device copies must not share an engine queue: a copy that waits for a spinning kernel would
# block the next rank's copy in (measured at copies above the runtime's blit threshold): copies as kernels on each
# rank's own stream. One process per GPU (the production layout) needs neither setting.
os.environ.setdefault("GPU_FORCE_BLIT_COPY_SIZE", "1048576")

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return graft.load_package()


@pytest.fixture(scope="session")
def fem(pkg):
    return pkg.fem


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle


def one(x, y):
    return 1.0 + 0 * x


def f_m1(x, y):
    return -1.0 + 0 * x


def u0734(x, y):
    return 0.734 + 0 * x


def a_example01(x, y):
    return 0.1 + 0.0001 * x * y          # Example01:33-35


def u3(x, y):
    return 3.0 + 0 * x


def lognormal_coeff(fem, points, seed=481456):
    """Config-3 coefficient: a = exp(g), g from the synthetic KL modes."""
    kl = fem.synthetic_kl(points)
    _, g = fem.draw(kl, np.random.default_rng(seed))
    return np.exp(g)


@pytest.fixture(scope="session")
def micro(fem):
    """~1.4 k free DoF, 2x2 subdomains, a = 1 (the micro case of SURVEY.md §7 step 1)."""
    return fem.build_schur_problem(40, 2, 2, one, f_m1, u0734)


@pytest.fixture(scope="session")
def toy(fem):
    """Config 1: N=100, 2x2 boxes, a = 1, f = -1, uexact = 0.734 (Example03:63-73). n_Γ = 195."""
    return fem.build_schur_problem(100, 2, 2, one, f_m1, u0734)


@pytest.fixture(scope="session")
def ragged(fem):
    """3x2 boxes on N=50 with a lognormal coefficient: unequal subdomains, cnt in {2,4}, odd n_Γd."""
    mesh = fem.get_mesh(50)
    return fem.build_schur_problem(50, 3, 2, lognormal_coeff(fem, mesh.points, 7), f_m1, u0734)


def lowest_eigvecs(orc_op, n, nev):
    """Stand-in for KrylovKit.eigsolve(..., :SR) (Example03:209): dense eigh of the assembled operator."""
    S = np.column_stack([orc_op(e) for e in np.eye(n)])
    w, V = np.linalg.eigh((S + S.T) / 2)
    return np.asfortranarray(V[:, :nev])


@pytest.fixture(scope="session")
def full(fem):
    """Config 3 / 4 at BASELINE.json's full size: N=1000 (996 004 free DoF), 4x2 subdomains, lognormal coefficient."""
    mesh = fem.get_mesh(1000)
    return fem.build_schur_problem(1000, 4, 2, lognormal_coeff(fem, mesh.points), f_m1, u0734)


@pytest.fixture(scope="session")
def ctx(pkg):
    """GPU context; only -m gpu tests request it."""
    return pkg.api.Context(0)


def unstructured_mesh(fem, N=28, nsec=6, seed=11):
    """A mesh + partition shaped like the output of the reference's Triangle + METIS pipeline as far as any code can
    tell: jittered interior nodes, node and element numbers randomly permuted, vertex order rotated per element, and
    `nsec` pie-slice subdomains that all meet at one interior node (multiplicity nsec there — box partitions stop at 4).
    Returns (Mesh, epart, npart), everything 0-based."""
    rng = np.random.default_rng(seed)
    m = fem.get_mesh(N)
    nnode, nel = m.points.shape[1], m.cells.shape[1]
    h = 1.0 / (N - 1)
    pts = m.points.copy()
    interior = m.point_marker == 0
    pts[:, interior] += rng.uniform(-0.22 * h, 0.22 * h, size=(2, int(interior.sum())))
    cx, cy = pts[:, (N // 2) * N + N // 2]                       # the hub: an interior NODE, so every slice touches it
    cen = pts[:, m.cells].mean(axis=1)                           # (2, nel) centroids
    ang = np.arctan2(cen[1] - cy, cen[0] - cx) + 0.4
    epart = np.floor((ang % (2 * np.pi)) / (2 * np.pi / nsec)).astype(np.int64) % nsec
    # renumber nodes and elements, rotate vertex order (orientation kept; the neighbour table rotates with it)
    pn, pe = rng.permutation(nnode), rng.permutation(nel)       # new id of old node / new position -> old element
    inv_e = np.empty(nel, dtype=np.int64); inv_e[pe] = np.arange(nel)
    cells, nb = pn[m.cells][:, pe], m.cell_neighbors[:, pe]
    nb = np.where(nb < 0, -1, inv_e[np.maximum(nb, 0)])
    rot = rng.integers(0, 3, nel)
    idx = (np.arange(3)[:, None] + rot[None, :]) % 3
    cells, nb = np.take_along_axis(cells, idx, 0), np.take_along_axis(nb, idx, 0)
    points = np.empty_like(pts); points[:, pn] = pts
    marker = np.empty_like(m.point_marker); marker[pn] = m.point_marker
    epart = epart[pe]
    npart = np.full(nnode, -1, dtype=np.int64)
    for k in range(3):
        npart[cells[k]] = epart                                  # any owning element's subdomain (mpmetis gives one of them)
    return fem.Mesh(cells, points, marker, nb, N), epart, npart

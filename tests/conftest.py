import os
import sys

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
def ctx(pkg):
    """GPU context; only -m gpu tests request it."""
    return pkg.api.Context(0)

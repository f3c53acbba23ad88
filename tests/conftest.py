import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    return load


def golden_names():
    """Fixtures whose solves ran on the matrix as assembled (the `perm_*` ones re-order it first)."""
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz") and not f.startswith(("perm_", "eig_", "comp_")))


def eig_golden_names():
    """Fixtures of lanczos / generalized_lanczos (time-seeded start vector kept in the fixture)."""
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz") and f.startswith("eig_"))


def perm_golden_names(ell=False):
    """Fixtures of the re-ordering path: permutations.f90 + symmetric permutation, solves on the
    permuted matrix (CSR ones by default, the ELLPACK ones with ell=True)."""
    return sorted(f[:-4] for f in os.listdir(GOLDEN)
                  if f.endswith(".npz") and f.startswith("perm_") and (("_ell_" in f) == ell))


def comp_golden_names():
    """Fixtures of the composite `sparse_matrix` (2 x 2 blocks; products and solves on the composite)."""
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz") and f.startswith("comp_"))

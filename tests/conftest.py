import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
TOL = 1e-12   # north_star: max|x - x_ref| / max|x_ref| <= 1e-12 per output array


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible and -m gpu was not asked for."""
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(tag):
        return np.load(os.path.join(GOLDEN, "ref_%s.npz" % tag))
    return load


# tag -> (trunc, ix, iy, kx): the stock builds and the other level counts (oracle/build_ref.sh variants)
VARIANTS = {"t30": (30, 96, 24, 8), "t63": (63, 192, 48, 8), "t30k5": (30, 96, 24, 5), "t30k7": (30, 96, 24, 7),
            "t63k16": (63, 192, 48, 16), "t30k20": (30, 96, 24, 20)}


@pytest.fixture(scope="session")
def oracle_factory():
    from oracle.pyoracle import Oracle, build
    import synth
    build()
    cache = {}

    def get(tag):
        if tag not in cache:
            cache[tag] = Oracle(*VARIANTS[tag])
            if tag in synth.SIGMA_SETS:               # level counts the reference has no sigma set for (geometry.f90:42-48)
                cache[tag].set_sigma(synth.SIGMA_SETS[tag])
        return cache[tag]
    return get

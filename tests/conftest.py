import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle

    return Oracle()


@pytest.fixture(scope="session")
def ref():
    from oracle.oracle import Ref

    if not Ref.available():
        pytest.skip("oracle/_ref/libsvs_ref.so not available (no reference tree / no AVX-512)")
    return Ref()


def deep_like(n, d, latent, seed, normalize=True, sigma=0.1, proj_seed=7):
    """SURVEY 8d generator G(latent, sigma): x = normalize(z A + sigma eps)."""
    a = np.random.default_rng(proj_seed).standard_normal((latent, d)).astype(np.float32)
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, latent)).astype(np.float32) @ a + sigma * rng.standard_normal((n, d)).astype(np.float32)
    if normalize:
        x /= np.linalg.norm(x, axis=1, keepdims=True)
    return np.ascontiguousarray(x, dtype=np.float32)


@pytest.fixture(scope="session")
def small_world(oracle):
    """A 1500 x 96 deep-like set, an oracle-built Vamana graph over it and 64 queries (shared by several tests)."""
    from oracle.oracle import L2

    x = deep_like(1500, 96, 24, seed=1)
    q = deep_like(64, 96, 24, seed=2)
    graph, entry = oracle.build_vamana(L2, x, R=32, L=48)
    return dict(x=x, q=q, graph=graph, entry=entry)

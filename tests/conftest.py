import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
MODES = ["global", "local", "semiglobal_both", "semiglobal_one", "semiglobal_two"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Load order matters where torch and libpraline_dp.so share a process: torch brings its own HIP runtime, and
    # torch.cuda.is_available() turns False when another copy (the system one libpraline_dp.so links) was loaded first.
    # The multi-GPU tests use torch, so it goes first here - as in bench.py and in any torch.distributed program, where
    # the process group exists before the library is touched.
    try:
        import torch  # noqa: F401
    except Exception:
        pass


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a box without a HIP device skips the gpu tests instead of failing them one by one
    (`-m gpu` on the GPU box runs them; the library itself still fails loudly without a device)."""
    if not any("gpu" in item.keywords for item in items):
        return
    try:
        from praline_amd import native
        have = native.device_count() > 0
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def one_hot(values, A):
    """PlainTrack -> one-hot float32 profile (praline/component/align.py:164-170)."""
    p = np.zeros((len(values), A), dtype=np.float32)
    p[np.arange(len(values)), np.asarray(values, dtype=np.int64)] = 1.0
    return p


def synth_lengths(rng, N, mu):
    """L ~ round(Normal(mu, 0.1 mu)) clipped to [0.5 mu, 1.5 mu]."""
    return np.clip(np.rint(rng.normal(mu, 0.1 * mu, N)), 0.5 * mu, 1.5 * mu).astype(int)


def synth_profile(rng, L, A=27, hi=20):
    """SURVEY 8(d) C2 profile: one-hot x5 counts + 6 random extra residues with counts 1-3,
    row-normalised the way ProfileTrack.profile does (praline/container/sequence.py:200-202)."""
    counts = np.zeros((L, A), dtype=np.int64)
    res = rng.integers(0, hi, L)
    counts[np.arange(L), res] += 5
    for _ in range(6):
        idx = rng.integers(0, hi, L)
        counts[np.arange(L), idx] += rng.integers(1, 4, L)
    totals = np.array(counts.sum(axis=1), dtype=np.float32)
    return np.array(counts / totals[:, np.newaxis], dtype=np.float32), counts


@pytest.fixture(scope="session")
def bba():
    d = load_golden("bba0184_inputs.npz")
    return {"S": d["blosum62"], "seqs": [d["seq%d" % i] for i in range(5)],
            "motif": [d["motif%d" % i] for i in range(5)], "ss": [d["ss%d" % i] for i in range(5)],
            "motif_matrix": d["motif_matrix"], "ss_matrix": d["ss_matrix"]}

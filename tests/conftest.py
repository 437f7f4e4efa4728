import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Load tests/golden/<name>.npz; `points_ref == 'bundled'` resolves to bundled.npz's cloud."""
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    if "points_ref" in d:
        d["points"] = np.load(os.path.join(GOLDEN, "bundled.npz"))["pts"]
    if "seed" in d and "points" not in d and "n" in d:
        from trajectory_optimization_amd import synth
        pts = synth.make_cloud(int(d["n"]), seed=int(d["seed"]),
                               extent=(30.0, 30.0, 30.0) if name.startswith("frustum") else (40.0, 40.0, 4.0))
        if "centre" in d:
            pts = pts - d["centre"].astype(np.float32)
        if "dup" in d:   # exact duplicate rows: the generator's recipe (tests/golden/make_golden.py)
            pts, d["source_row"] = synth.with_duplicate_rows(pts, d["dup"], int(d["seed"]) + 1000)
            assert len(pts) == int(d["n_rows"])
        d["points"] = pts
    return d


def lowest_identical_rows(points, idx):
    """idx with every row replaced by the lowest-index row of `points` that has the same coordinates (ascending, unique): the
    rule the GPU hull reports duplicated rows by."""
    pts = np.ascontiguousarray(points)
    _, first, inverse = np.unique(pts.view([("", pts.dtype)] * 3).ravel(), return_index=True, return_inverse=True)
    return np.unique(first[inverse.ravel()[np.asarray(idx)]])


def rel_inf(a, b):
    """||a-b||_inf / ||b||_inf — the gradient parity measure of BASELINE.md §3."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    den = np.abs(b).max()
    return np.abs(a - b).max() / (den if den > 0 else 1.0)


@pytest.fixture(scope="session")
def golden():
    return load_golden

import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """name -> dict(opts, words, n_samples, sha256_*); made by tests/golden/make_golden.py
    from the reference's own compiled filter."""
    with open(os.path.join(GOLDEN_DIR, "golden_manifest.json")) as f:
        man = json.load(f)
    z = np.load(os.path.join(GOLDEN_DIR, "golden.npz"))
    out = {}
    for c in man["cases"]:
        d = dict(c)
        d["opts"] = tuple(c["opts"])
        if not c["hash_only"]:
            d["words"] = z[c["name"] + "/words"]
        out[c["name"]] = d
    return out


def golden_case_names(include_hash_only=False):
    with open(os.path.join(GOLDEN_DIR, "golden_manifest.json")) as f:
        man = json.load(f)
    return [c["name"] for c in man["cases"] if include_hash_only or not c["hash_only"]]


def delta_only(opts):
    """True when compression_opts selects the default [1,-1] prediction filter."""
    return len(opts) <= 2

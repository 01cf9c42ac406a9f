"""Shared pytest configuration.

* ``gpu`` marker: tests that need a real MI355X (run with ``-m gpu`` on the GPU box).
* Everything else runs on CPU only (``-m "not gpu"``) and never launches a kernel.
* ``oracle/`` (test infrastructure) is importable as ``oq_oracle`` from tests only.
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_sessionstart(session):
    """A GPU box that received the sources without the built library (a fresh clone: `*.so` is git-ignored) builds it in
    tree with hipcc before the first test, exactly as `__graft_entry__.build()` does; nothing is ever substituted for it."""
    if not _has_gpu():
        return
    from onnx_quantize_amd import _build
    if not os.path.exists(_build.LIB):
        print(f"[conftest] {_build.LIB} missing: building it with hipcc", file=sys.stderr)
        _build.build(verbose=True)


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def rng():
    # same seed as the reference's shared fixture (test/conftest.py:5-8)
    return np.random.default_rng(42)


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name))


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def synth_weight(kind: str, seed: int, k: int, n: int) -> np.ndarray:
    """Same recipes as tests/golden/make_golden.py::weight (inputs of the digest cases)."""
    r = np.random.default_rng(seed)
    if kind == "normal":
        return r.standard_normal((k, n), dtype=np.float32)
    if kind == "heavy":
        return r.standard_t(3, size=(k, n)).astype(np.float32)
    if kind == "zero_groups":
        w = r.standard_normal((k, n), dtype=np.float32)
        w[: k // 2, ::3] = 0.0
        w[:, 1] = 0.0
        return w
    raise ValueError(kind)

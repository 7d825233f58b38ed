"""Test configuration: registers the ``gpu`` marker and puts the product package directory
(``opensearch-neural-pre-train_amd/`` -- holds ``snx`` and the ``src`` mirror of the reference
interface) and the repo root (``oracle``) on sys.path."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "opensearch-neural-pre-train_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Collection order of the GPU files: the per-kernel, fp32-golden, whole-model and full-size PARITY tests first, the
# process-plumbing files (2-rank rehearsals, torchrun subprocesses, bench contract) last -- under the driver's `-x` a
# plumbing failure must not be able to hide the parity record.  Files not named keep their alphabetical place in between.
_ORDER = ["test_gpu_ops.py", "test_gpu_f32.py", "test_gpu_model.py", "test_gpu_parity_full.py"]
_LAST = ["test_gpu_dist.py", "test_gpu_bench_contract.py"]


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        name = os.path.basename(str(item.fspath))
        if name in _ORDER:
            return _ORDER.index(name)
        if name in _LAST:
            return 1000 + _LAST.index(name)
        return 500
    items.sort(key=rank)          # stable: the order inside a file is kept


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN

import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    import torch
    return torch.cuda.is_available()


def pytest_collection_modifyitems(config, items):
    # GPU tests never fall back to a CPU path: without a device they are skipped, with one they must load libsailor_hip.so
    if any("gpu" in item.keywords for item in items) and not _has_gpu():
        skip = pytest.mark.skip(reason="no HIP device in this container")
        for item in items:
            if "gpu" in item.keywords:
                item.add_marker(skip)


@pytest.fixture(scope="session")
def ctx():
    from sailor_amd.forward_plus import HipContext
    c = HipContext("cuda:0")
    yield c
    c.close()


def oracle_tile_rows(num_rows: int, fixed, whole_from_threads: int = 32):
    """The tile rows a full-size test holds against the oracle, as (first row, count) spans: the WHOLE frame when the host has at least
    `whole_from_threads` threads for the threaded checkers (oracle_light_cull_threads / oracle_shade_threads: the GPU box has 256, and the 4K frame
    takes seconds there; a host with fewer than 32 threads checks the fixed spans instead of spending a minute per frame -- ADVICE r03), otherwise the fixed spans given.  SAILOR_ORACLE_ROWS=r0:n[,r0:n...] overrides both (a failure on some other row is
    reproduced by naming it); nothing here depends on the date."""
    from oracle import oracle
    env = os.environ.get("SAILOR_ORACLE_ROWS")
    if env:
        return [tuple(int(v) for v in span.split(":")) for span in env.split(",")]
    if oracle.host_threads() >= whole_from_threads:
        return [(0, num_rows)]
    return list(fixed)

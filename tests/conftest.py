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


def daily_tile_row(num_rows: int, fixed: int, span: int = 1) -> int:
    """A tile row for the oracle-sampled part of a full-size test that changes from day to day (seeded by the date, so a run is
    reproducible on its day and prints what it used): the fixed row of the test stays, this one widens the coverage over time."""
    import datetime
    import random
    d = datetime.date.today()
    seed = int(os.environ.get("SAILOR_DAILY_SEED", d.year * 10000 + d.month * 100 + d.day))  # (override: sweep other rows on demand)
    rng = random.Random(seed)
    r = rng.randrange(0, num_rows - span + 1)
    if abs(r - fixed) < span:
        r = (fixed + span + 7) % (num_rows - span + 1)
    print(f"[daily oracle row] {d.isoformat()} (seed {seed}): tile rows {r}..{r + span - 1} of {num_rows}")
    return r

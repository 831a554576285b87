"""A bounded slice of the randomised parity sweeps inside `pytest -m gpu`: the same cases as scripts/fuzz_parity.py and
scripts/fuzz_csm_ecs.py (tests/fuzz_cases.py), fixed seeds (300 K1 + K2, 100 K3, 100 K4 cases: under a minute in all), alternating between the plain and the prepared-lights entry points.  The seeds are FIXED so that the gate is hermetic (the
same commit gives the same verdict on any day); SAILOR_FUZZ_SEEDS=s1,s2,... adds further seeds on demand (a nightly job can pass the date), and the
scripts run thousands of cases through gpurun.  A failure's message names seed and case: `python scripts/fuzz_parity.py <cases> <seed> <case>` replays it."""
import os

import numpy as np
import pytest

import fuzz_cases

pytestmark = pytest.mark.gpu

EXTRA = [int(s) for s in os.environ.get("SAILOR_FUZZ_SEEDS", "").split(",") if s.strip()]
K1K2_CASES = int(os.environ.get("SAILOR_FUZZ_K1K2_CASES", "300"))
K3_CASES = int(os.environ.get("SAILOR_FUZZ_K3_CASES", "100"))
K4_CASES = int(os.environ.get("SAILOR_FUZZ_K4_CASES", "100"))


@pytest.mark.parametrize("seed", [20250301] + EXTRA)
def test_random_frames_through_every_cull_path_and_the_shade(ctx, seed):
    rng = np.random.default_rng(seed)
    worst = 0.0
    for c in range(K1K2_CASES):
        try:
            worst = max(worst, fuzz_cases.k1k2_case(ctx, rng, c))
        except AssertionError as e:
            raise AssertionError(f"seed {seed}, case {c} (replay: scripts/fuzz_parity.py {c + 1} {seed} {c}): {e}") from e
    assert worst <= 1e-4
    print(f"[fuzz K1+K2] seed {seed}: {K1K2_CASES} cases, worst relative radiance error {worst:.2e}")


@pytest.mark.parametrize("seed", [20250302] + EXTRA)
def test_random_shadowed_frames(ctx, seed):
    rng = np.random.default_rng(seed)
    worst = 0.0
    for c in range(K3_CASES):
        try:
            worst = max(worst, fuzz_cases.k3_case(ctx, rng, c))
        except AssertionError as e:
            raise AssertionError(f"seed {seed}, K3 case {c}: {e}") from e
    assert worst <= 1e-4
    print(f"[fuzz K3] seed {seed}: {K3_CASES} cases, worst relative radiance error {worst:.2e}")


@pytest.mark.parametrize("seed", [20250303] + EXTRA)
def test_random_hierarchies_through_the_ecs_sweep(ctx, seed):
    rng = np.random.default_rng(seed)
    for c in range(K4_CASES):
        try:
            fuzz_cases.k4_case(ctx, rng, c)
        except AssertionError as e:
            raise AssertionError(f"seed {seed}, K4 case {c}: {e}") from e


def test_random_caster_draws_and_indirect_draw_compactions():
    """Round 6: a bounded slice of scripts/fuzz_raster_meshcull.py inside the suite (60 rasteriser cases -- every fifth a draw of >= 4 096 instances on a larger
    map: chunked launches, the instance test, the giant triangles' queue and its overflow, dependent passes on a coarse depth that starts from zero -- and 60
    compaction cases), depth buffers and instance buffers bit for bit; a child process, as the script is run through gpurun by the thousand."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    p = subprocess.run([sys.executable, "scripts/fuzz_raster_meshcull.py", "60", "20250304"], cwd=root, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-2000:]
    assert "raster fuzz ok: 60 cases" in p.stdout and "mesh cull fuzz ok: 60 cases" in p.stdout, p.stdout[-500:]

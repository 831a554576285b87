"""Regenerates tests/golden/*.npz: frozen synthetic inputs + the CPU oracle's outputs for them.

The reference (aantropov/Sailor) has no tests, golden vectors or fixtures for this path and cannot be built or run here
(SURVEY.md 4, 8c), so these are NOT reference outputs: they pin (a) the synthetic generator -- inputs must regenerate
byte-identically -- and (b) the oracle's results on them, so a change of either is caught.  Run from the repo root:
    python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import oracle  # noqa: E402
from sailor_amd import synth  # noqa: E402

OUT = Path(__file__).resolve().parent


def make(name: str) -> None:
    f = synth.make_frame(name)
    W, H = f.cam.width, f.cam.height
    g, idx, cnt = oracle.light_cull(f.cam.frame, W, H, f.lights, f.depth, want_counts=True)
    csm = None
    extra = {}
    if f.shadows is not None:
        csm, _keep = oracle.make_csm(f.shadows.lights_matrices, f.shadows.maps)
        extra["lights_matrices"] = f.shadows.lights_matrices
        for k in range(4):
            extra[f"shadow_map{k}"] = f.shadows.maps[k]
    rad = oracle.shade(f.cam.frame, W, H, f.surface, f.lights, g, idx, csm)
    np.savez_compressed(OUT / f"{name}.npz",
                        frame_ubo=np.frombuffer(bytes(f.cam.frame), np.uint8), depth=f.depth,
                        lights=np.frombuffer(f.lights.tobytes(), np.uint8), surface=f.surface,
                        grid=g, indices=idx[: 1 + int(idx[0])], passing=cnt, radiance=rad, **extra)
    print(f"{name}: {W}x{H}, {len(f.lights)} lights, sum(num) = {int(idx[0])}, mean list {g[:, 1].mean():.2f}, "
          f"tiles >128 candidates {(cnt > 128).sum()}, >196 {(cnt > 196).sum()}, lit pixels {(rad[..., :3].sum(-1) > 0).mean():.3f}")


def make_depth() -> None:
    """LinearizeDepth: raw reversed-Z attachment of the tiny frame (6 % sky blocks, plus hand-picked
    edge values in the first row) and the oracle's linearisation of it."""
    f = synth.make_frame("tiny", with_surface=False)
    raw = synth.make_raw_depth(f.depth, 1.0, sky_fraction=0.06)
    raw[0, :8] = np.array([1.0, 0.5, 1e-4, 3.0e-39, 0.0, 0.99999994, 1.1754944e-38, 2.5e-5], np.float32)
    lin = oracle.linearize_depth(1.0, raw)
    np.savez_compressed(OUT / "tiny_depth.npz", z_near=np.float32(1.0), raw=raw, linear=lin)
    print(f"tiny_depth: {raw.shape[1]}x{raw.shape[0]}, {(raw == 0).mean():.3f} sky, linear range [{lin[np.isfinite(lin)].min():.3f}, {lin[np.isfinite(lin)].max():.3e}]")


# what the stored radiance was sampled with -- NOT the reference's samplers ("what's weak" 11)
SAMPLER_NOTE = ("canonical sampler of oracle/sailor_oracle.c, self-defined: fp32 texels, cube face and (s, t) by the Vulkan major-axis table (ties z over y over x), "
                "bilinear inside the face with clamp-to-edge, linear between the two nearest mips, lod clamped to [0, levels - 1].  The reference samples RGBA16F "
                "images through driver-defined Vulkan samplers; its cubemap bakes accumulate sequentially, the GPU's by a fixed tree (tolerance-checked).")


def make_ibl() -> None:
    """Ambient term: the oracle's ComputeBrdfLut table (32x32) and the tiny frame shaded with the synthetic
    IBL set (the cubemaps / AO regenerate from the frozen generator, so only the table and the result are stored)."""
    f = synth.make_frame("tiny")
    W, H = f.cam.width, f.cam.height
    lut = oracle.compute_brdf_lut(32, 32)
    ibl = synth.make_ibl_set(W, H, lut)
    g, idx, _ = oracle.light_cull(f.cam.frame, W, H, f.lights, f.depth)
    oibl, _keep = oracle.make_ibl(ibl.irradiance, ibl.env_chain, ibl.env_size, ibl.env_levels, ibl.brdf_lut, ibl.ao)
    rad = oracle.shade(f.cam.frame, W, H, f.surface, f.lights, g, idx, ibl=oibl)
    np.savez_compressed(OUT / "tiny_ibl.npz", brdf_lut=lut, radiance=rad, env_checksum=np.float64(ibl.env_chain.astype(np.float64).sum()),
                        irr_checksum=np.float64(ibl.irradiance.astype(np.float64).sum()), ao_checksum=np.float64(ibl.ao.astype(np.float64).sum()),
                        sampler=np.array(SAMPLER_NOTE))
    print(f"tiny_ibl: lut range [{lut.min():.4f}, {lut.max():.4f}], mean radiance {rad[..., :3].mean():.3f}")


def make_blur() -> None:
    """EVSM blur: cascade 0 of the tiny_csm shadow set (64x64 RGBA32F) blurred with ShadowCascadeBlur[0] = (2, 5)."""
    f = synth.make_frame("tiny_csm", with_surface=False)
    m = np.ascontiguousarray(f.shadows.maps[0])
    np.savez_compressed(OUT / "tiny_blur.npz", radii=np.array([2, 5], np.int32), blurred=oracle.evsm_blur(m, 2, 5))
    print(f"tiny_blur: {m.shape}")


def make_mesh_cull() -> None:
    """GPU-culling Dispatch: 3 000 instances in 24 draws behind 9 untouched records, a 96 x 54 raw depth image and its 96 x 96
    min pyramid; the oracle's pyramid, the instance / indirect buffers after frustum-only and after frustum + occlusion culling with compaction.
    Inputs regenerate from the frozen generator; outputs are stored (instances as their 24 uint32 words)."""
    cam = synth.make_camera(640, 360)
    s = synth.make_instance_set(3000, 24, first_instance=9)
    lin = synth.make_linear_depth(96, 54, 5, d_min=200.0, d_max=2500.0)
    raw = synth.make_raw_depth(lin, cam.frame.cameraZNearZFar[0])
    pyr = oracle.hiz_build(raw, 96, 96, 7)
    fi, fb = oracle.mesh_cull_compact(cam.frame, s.instances, 3000, 9, s.batches)
    oi, ob = oracle.mesh_cull_compact(cam.frame, s.instances, 3000, 9, s.batches, hiz=(pyr, 96, 96, 7))
    np.savez_compressed(OUT / "tiny_mesh_cull.npz", pyramid=pyr, frustum_instances=fi.view(np.uint32).reshape(-1, 24), frustum_batches=fb,
                        occlusion_instances=oi.view(np.uint32).reshape(-1, 24), occlusion_batches=ob)
    print(f"tiny_mesh_cull: kept {int(fb[:, 1].sum())} of 3000 after the frustum test, {int(ob[:, 1].sum())} with occlusion")


def make_prefilter() -> None:
    """IBL bake: the synthetic 16 x 16 x 6 sky with 5 mips -> pre-filtered environment cube (every mip) and a 2 x 2 x 6
    irradiance cube, by the oracle."""
    sky = synth.make_ibl_set(16, 16, np.zeros((2, 2, 2), np.float32), env_size=16, with_ao=False)
    env = oracle.prefilter_env_map(sky.env_chain, 16, sky.env_levels)
    irr = oracle.compute_irradiance_map(env, 16, sky.env_levels, 2)
    np.savez_compressed(OUT / "tiny_prefilter.npz", env=env, irradiance=irr, raw_checksum=np.float64(sky.env_chain.astype(np.float64).sum()))
    print(f"tiny_prefilter: env {env.size // 4} texels, irradiance mean {irr[..., :3].mean():.4f}")


def make_raster() -> None:
    """The canonical depth rasteriser: 1 500 triangles around the eye through a reversed-Z perspective matrix -- near-plane
    cuts, back-face culling on and off -- at 96 x 64."""
    pos, idx = synth.make_triangle_soup(1500)
    P = synth.perspective_reversed_z(96, 64)
    one = np.eye(4, dtype=np.float32).reshape(1, 16)
    both = oracle.raster_depth(P, pos, idx, one, 96, 64)
    front = oracle.raster_depth(P, pos, idx, one, 96, 64, cull_back=True)
    np.savez_compressed(OUT / "tiny_raster.npz", both=both, front=front, soup_checksum=np.float64(pos.astype(np.float64).sum()))
    print(f"tiny_raster: covered {float((both > 0).mean()):.3f} / {float((front > 0).mean()):.3f} (front faces only)")


if __name__ == "__main__":
    for n in ("tiny", "tiny_csm"):
        make(n)
    make_depth()
    make_ibl()
    make_blur()
    make_mesh_cull()
    make_prefilter()
    make_raster()

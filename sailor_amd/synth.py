"""Deterministic synthetic frames for the Forward+ path (SURVEY.md 8d "Synthetic frame generator").

RNG = counter-based SplitMix64 -> 24-bit uniforms u = (x >> 40) * 2^-24; seed 20250114; one stream per input class
(camera 0, depth 1, surface 2, lights 3, shadow 4, entities 5).  Everything is plain numpy: the generator feeds the HIP
path, the oracle and the benchmarks with the SAME bytes; it performs none of the path's arithmetic.

Defaults harvested from the reference: camera at (0,150,0), identity rotation, fov 90, zNear 1, zFar 20000
(Content/Editor.world:5-9,29-31); light attenuation (1, 0.022, 0.0019) and cut-off (30, 45) degrees
(ECS/LightingECS.h:24-26); integer light intensities 0..255 (Components/TestComponent.cpp:117); the directional
light's intensity from Content/Editor.world:123-126 (its rotation: see directional_rotation()).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import host

SEED = 20250114
STREAM_CAMERA, STREAM_DEPTH, STREAM_SURFACE, STREAM_LIGHTS, STREAM_SHADOW, STREAM_ENTITIES = range(6)

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _mix(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _stream_base(seed: int, stream: int) -> np.uint64:
    return _mix(np.array([(seed * 0x100000001B3 + stream * 0xD1B54A32D192ED03 + 0x632BE59BD9B4E019) & 0xFFFFFFFFFFFFFFFF], np.uint64))[0]


def uniforms(stream: int, count: int, offset: int = 0, seed: int = SEED) -> np.ndarray:
    """count uniforms in [0,1) (float32, exact 24-bit values): element k is output k+offset of the stream."""
    k = np.arange(offset + 1, offset + count + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = _mix(_stream_base(seed, stream) + k * _GOLDEN)
    return ((x >> np.uint64(40)).astype(np.float32)) * np.float32(2.0 ** -24)


# ---------------------------------------------------------------------------------------------------------------
@dataclass
class Camera:
    world: np.ndarray            # float32[16], column-major
    fov: float
    z_near: float
    z_far: float
    width: int
    height: int
    frame: object = None         # _lib.UboFrameData

    @property
    def aspect(self) -> float:
        return float(np.float32(self.width) / np.float32(self.height))


def make_camera(width: int, height: int, fov: float = 90.0, z_near: float = 1.0, z_far: float = 20000.0) -> Camera:
    world = host.transform_matrix([0.0, 150.0, 0.0, 1.0], [0.0, 0.0, 0.0, 1.0], [1.0, 1.0, 1.0, 1.0])
    cam = Camera(world=world, fov=fov, z_near=z_near, z_far=z_far, width=width, height=height)
    cam.frame = host.fill_frame_data(world, fov, z_near, z_far, width, height)
    return cam


def _value_noise(width: int, height: int, cell: int, stream: int, offset: int, seed: int) -> np.ndarray:
    """bilinear value noise in [0,1), float64[height, width]."""
    gw, gh = width // cell + 2, height // cell + 2
    lattice = uniforms(stream, gw * gh, offset, seed).astype(np.float64).reshape(gh, gw)
    xs = (np.arange(width) + 0.5) / cell
    ys = (np.arange(height) + 0.5) / cell
    x0 = np.floor(xs).astype(np.int64); fx = xs - x0
    y0 = np.floor(ys).astype(np.int64); fy = ys - y0
    a = lattice[np.ix_(y0, x0)]; b = lattice[np.ix_(y0, x0 + 1)]
    c = lattice[np.ix_(y0 + 1, x0)]; d = lattice[np.ix_(y0 + 1, x0 + 1)]
    top = a * (1 - fx)[None, :] + b * fx[None, :]
    bot = c * (1 - fx)[None, :] + d * fx[None, :]
    return top * (1 - fy)[:, None] + bot * fy[:, None]


def make_linear_depth(width: int, height: int, seed: int = SEED, d_min: float = 10.0, d_max: float = 3000.0) -> np.ndarray:
    """The `LinearDepth` render target (R32F positive view distance, LinearizeDepth.shader:69-73): row 0 = top.

    Value noise on a 64-px lattice mapped log-uniformly to [d_min, d_max]; 5 % of 16x16 blocks (offset by 8 px so
    that they straddle light tiles) are overwritten by a second, independent layer => depth discontinuities
    inside tiles, so the min/max window of the cull matters."""
    t = _value_noise(width, height, 64, STREAM_DEPTH, 0, seed)
    depth = d_min * (d_max / d_min) ** t
    t2 = _value_noise(width, height, 64, STREAM_DEPTH, 1 << 20, seed)
    depth2 = d_min * (d_max / d_min) ** t2
    bw, bh = (width + 8) // 16 + 1, (height + 8) // 16 + 1
    pick = uniforms(STREAM_DEPTH, bw * bh, 1 << 21, seed).reshape(bh, bw) < 0.05
    by = (np.arange(height) + 8) // 16
    bx = (np.arange(width) + 8) // 16
    mask = pick[np.ix_(by, bx)]
    depth = np.where(mask, depth2, depth)
    return np.ascontiguousarray(depth.astype(np.float32))


def make_raw_depth(linear: np.ndarray, z_near: float = 1.0, sky_fraction: float = 0.0, seed: int = SEED) -> np.ndarray:
    """A reversed-Z, infinite-far-plane depth attachment that LinearizeDepth.shader:70 maps back to (about) `linear`:
    raw = zNear / linear in float32 (so linearising it is a rounding away from `linear`, not the identity).  With
    `sky_fraction` > 0 that share of 16x16 blocks is cleared to 0.0 = nothing drawn (linear depth +inf)."""
    raw = (np.float32(z_near) / linear.astype(np.float32)).astype(np.float32)
    if sky_fraction > 0.0:
        H, W = linear.shape
        bw, bh = (W + 15) // 16, (H + 15) // 16
        pick = uniforms(STREAM_DEPTH, bw * bh, 1 << 22, seed).reshape(bh, bw) < sky_fraction
        raw = np.where(pick[np.ix_(np.arange(H) // 16, np.arange(W) // 16)], np.float32(0.0), raw)
    return np.ascontiguousarray(raw, np.float32)


def pixel_rays(cam: Camera) -> tuple[np.ndarray, np.ndarray]:
    """Appendix D: pixel (px, py) (py = 0 top) has NDC (2(px+.5)/W - 1, 1 - 2(py+.5)/H); view ray (ndc.x/P00, ndc.y/P11, -1)."""
    proj = np.frombuffer(bytes(cam.frame.projection), np.float32)
    p00, p11 = proj[0], proj[5]
    W, H = cam.width, cam.height
    ndc_x = (2.0 * (np.arange(W, dtype=np.float32) + np.float32(0.5)) / np.float32(W) - 1.0).astype(np.float32)
    ndc_y = (1.0 - 2.0 * (np.arange(H, dtype=np.float32) + np.float32(0.5)) / np.float32(H)).astype(np.float32)
    return (ndc_x / p00).astype(np.float32), (ndc_y / p11).astype(np.float32)


def make_surface(cam: Camera, depth: np.ndarray, seed: int = SEED, row_begin: int = 0, row_end: int | None = None) -> np.ndarray:
    """Surface planes float32[3, rows, W, 4] for framebuffer rows [row_begin, row_end):
    P0 = (worldPos.xyz, albedo.a in U[0.5,1]); P1 = (normalize(-viewDir + 0.8*noise), roughness in U[0.05,1]);
    P2 = (albedo.rgb in U[0.05,1]^3, metallic: 0 w.p. 0.8 else U[0,1])."""
    W, H = cam.width, cam.height
    row_end = H if row_end is None else row_end
    rows = row_end - row_begin
    rx, ry = pixel_rays(cam)
    world = cam.world.reshape(4, 4)  # world[c] = column c
    d = depth[row_begin:row_end].astype(np.float32)
    vx = rx[None, :] * d
    vy = ry[row_begin:row_end, None] * d
    vz = -d
    out = np.empty((3, rows, W, 4), np.float32)
    for i in range(3):  # worldPos = cameraWorld * (view, 1)
        out[0, :, :, i] = world[0, i] * vx + world[1, i] * vy + world[2, i] * vz + world[3, i]
    u = uniforms(STREAM_SURFACE, rows * W * 10, row_begin * W * 10, seed).reshape(rows, W, 10)
    out[0, :, :, 3] = 0.5 + 0.5 * u[:, :, 0]
    cam_pos = world[3, :3]
    vd = out[0, :, :, :3] - cam_pos[None, None, :]
    vd /= np.sqrt((vd * vd).sum(-1, keepdims=True))
    n = -vd + np.float32(0.8) * (2.0 * u[:, :, 1:4] - 1.0)
    n /= np.sqrt((n * n).sum(-1, keepdims=True))
    out[1, :, :, :3] = n
    out[1, :, :, 3] = 0.05 + 0.95 * u[:, :, 4]
    out[2, :, :, :3] = 0.05 + 0.95 * u[:, :, 5:8]
    out[2, :, :, 3] = np.where(u[:, :, 8] < 0.8, np.float32(0.0), u[:, :, 9])
    return out


@dataclass
class LightSetConfig:
    count: int
    spot_fraction: float = 0.0        # C2: 0, C3/C5: 0.25
    radius_scale: float = 1.0         # "s" of SURVEY 8d, frozen per config below
    cluster_lights: int = 0           # dense cluster(s) guaranteeing tiles with > 196 candidates
    cluster_count: int = 1
    cluster_spread: float = 3.0       # cluster half-width in tiles
    cluster_radius: tuple = (1.7, 3.4)  # cluster light radii in tiles
    directional_first: bool = False   # C4: light 0 = directional, EVSM
    d_min: float = 10.0
    d_max: float = 3000.0


def make_lights(cam: Camera, depth: np.ndarray, cfg: LightSetConfig, seed: int = SEED) -> np.ndarray:
    """LightShaderData records (host.LIGHT_DTYPE).  View depth log-uniform in [d_min, d_max]; x/y uniform in the
    frustum cross-section x 1.1; r = z * (2/240) * U[2,10] * s."""
    N = cfg.count
    u = uniforms(STREAM_LIGHTS, N * 12, 0, seed).reshape(N, 12).astype(np.float64)
    z = cfg.d_min * (cfg.d_max / cfg.d_min) ** u[:, 0]
    tan_half = np.tan(np.radians(cam.fov) * 0.5)
    x = (2 * u[:, 1] - 1) * 1.1 * z * tan_half * cam.aspect
    y = (2 * u[:, 2] - 1) * 1.1 * z * tan_half
    r = z * (2.0 / 240.0) * (2 + 8 * u[:, 3]) * cfg.radius_scale

    if cfg.cluster_lights > 0:
        # clusters hug the visible surface: every cluster light sits on the depth image at its own pixel, so the
        # tiles around a cluster centre see far more than 196 candidates (Appendix A's ">196" regime)
        W, H = cam.width, cam.height
        rx, ry = pixel_rays(cam)
        per = cfg.cluster_lights // cfg.cluster_count
        cu = uniforms(STREAM_LIGHTS, cfg.cluster_count * 2 + cfg.cluster_lights * 4, N * 12, seed).astype(np.float64)
        half = cfg.cluster_spread * 16.0                  # half-width in pixels (cluster_spread is in tiles)
        tile_w = cam.aspect * tan_half * 32.0 / W         # view-space width of one tile at unit depth
        for c in range(cfg.cluster_count):
            cx = (0.15 + 0.7 * cu[2 * c]) * W
            cy = (0.15 + 0.7 * cu[2 * c + 1]) * H
            idx = (np.arange(per) * (N // max(per, 1)) + c * 7 + 3) % N
            o = cfg.cluster_count * 2 + c * per * 4
            j = cu[o:o + per * 4].reshape(per, 4)
            px = np.clip((cx + (2 * j[:, 0] - 1) * half).astype(np.int64), 0, W - 1)
            py = np.clip((cy + (2 * j[:, 1] - 1) * half).astype(np.int64), 0, H - 1)
            dz = depth[py, px].astype(np.float64) * (0.98 + 0.04 * j[:, 2])
            z[idx] = dz
            x[idx] = rx[px] * dz
            y[idx] = ry[py] * dz
            r[idx] = dz * tile_w * (cfg.cluster_radius[0] + (cfg.cluster_radius[1] - cfg.cluster_radius[0]) * j[:, 3])

    world = cam.world.reshape(4, 4).astype(np.float64)
    lights = np.zeros(N, host.LIGHT_DTYPE)
    pos = np.empty((N, 3))
    for i in range(3):
        pos[:, i] = world[0, i] * x + world[1, i] * y + world[2, i] * (-z) + world[3, i]
    lights["worldPosition"] = pos.astype(np.float32)
    lights["bounds"] = np.repeat(r.astype(np.float32)[:, None], 3, axis=1)
    lights["intensity"] = np.floor(u[:, 4:7] * 256.0).astype(np.float32)
    lights["attenuation"] = np.array([1.0, 0.022, 0.0019], np.float32)
    c_in, c_out = host.cutoff_cosines(30.0, 45.0)
    lights["cutOff"] = np.array([c_in, c_out], np.float32)
    # random unit direction
    zz = 2 * u[:, 7] - 1
    phi = 2 * np.pi * u[:, 8]
    rr = np.sqrt(np.maximum(0.0, 1 - zz * zz))
    lights["direction"] = np.stack([rr * np.cos(phi), rr * np.sin(phi), zz], 1).astype(np.float32)
    lights["type"] = np.where(u[:, 9] < cfg.spot_fraction, host.LIGHT_SPOT, host.LIGHT_POINT).astype(np.uint32)
    lights["shadowType"] = host.SHADOW_PCF
    if cfg.directional_first and N > 0:
        q = directional_rotation().astype(np.float64)
        lights["type"][0] = host.LIGHT_DIRECTIONAL
        lights["shadowType"][0] = host.SHADOW_EVSM
        lights["direction"][0] = directional_forward(q).astype(np.float32)
        lights["intensity"][0] = np.array([17.0, 17.0, 17.0], np.float32)  # Editor.world:123-126
        lights["worldPosition"][0] = 0.0
    return lights


def directional_rotation() -> np.ndarray:
    """Unit quaternion (x,y,z,w) of the directional light: yaw 25 deg about +Y, then pitch -50 deg about +X, i.e. the
    light looks into the camera's frustum and down.  (The quaternion serialised at Content/Editor.world:110-114 is not
    unit length and, fed through the reference's own CSM math with the default camera, puts every visible fragment at
    light-clip z < 0 where Lighting.glsl:248-252,269-274 return 1 -- no pixel would exercise the shadow lookups.)"""
    yaw, pitch = np.radians(25.0), np.radians(-50.0)
    qy = np.array([0.0, np.sin(yaw / 2), 0.0, np.cos(yaw / 2)])
    qx = np.array([np.sin(pitch / 2), 0.0, 0.0, np.cos(pitch / 2)])
    ax, ay, az, aw = qy
    bx, by, bz, bw = qx
    q = np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                  aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])
    return (q / np.linalg.norm(q)).astype(np.float32)


def directional_forward(q) -> np.ndarray:
    """glm::rotate(q, vec3_Forward = (0,0,-1)) (Math/Transform.cpp:74, Math/Math.h:20)."""
    x, y, z, w = [float(v) for v in q]
    # third column of the rotation matrix, negated
    return -np.array([2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)])


@dataclass
class IblSet:
    irradiance: np.ndarray     # float32[6, S, S, 4]   g_irradianceCubemap (one level)
    env_chain: np.ndarray      # float32 flat mip chain of g_envCubemap, level-major, each level [6, s, s, 4]
    env_size: int
    env_levels: int
    brdf_lut: np.ndarray       # float32[H, W, 2]      g_brdfSampler (ComputeBrdfLut.shader)
    ao: np.ndarray | None      # float32[H, W]         g_aoSampler target, or None


def _cube_dirs(size: int) -> np.ndarray:
    """unit direction of every texel centre, [6, size, size, 3], face / (s, t) convention of the Vulkan major-axis table"""
    c = (np.arange(size, dtype=np.float64) + 0.5) / size * 2.0 - 1.0
    sc, tc = np.meshgrid(c, c)  # tc varies along rows
    one = np.ones_like(sc)
    faces = [(one, -tc, -sc), (-one, -tc, sc), (sc, one, tc), (sc, -one, -tc), (sc, -tc, one), (-sc, -tc, -one)]
    d = np.stack([np.stack(f, -1) for f in faces], 0)
    return d / np.linalg.norm(d, axis=-1, keepdims=True)


def make_ibl_set(width: int, height: int, lut, env_size: int = 64, irr_size: int = 16, seed: int = SEED, with_ao: bool = True) -> IblSet:
    """Analytic sky: a sun lobe + horizon gradient + a few coloured blobs, evaluated at texel centres; mips by 2x2 box filter;
    'irradiance' = a low-frequency version of the same sky (a stand-in for ComputeIrradianceMap.shader's output).  `lut` is the
    BRDF look-up table (float32[H, W, 2]) -- pass oracle.compute_brdf_lut(...) or the GPU's.  AO = value noise in [0.3, 1]."""
    u = uniforms(STREAM_SHADOW, 64, 1 << 27, seed).astype(np.float64)

    def sky(d, sharp):
        sun = np.array([0.3, 0.8, 0.52]); sun /= np.linalg.norm(sun)
        out = np.zeros(d.shape[:-1] + (4,), np.float64)
        up = d[..., 1] * 0.5 + 0.5
        out[..., 0] = 0.25 + 0.5 * up; out[..., 1] = 0.35 + 0.5 * up; out[..., 2] = 0.55 + 0.4 * up
        lobe = np.clip((d * sun).sum(-1), 0, 1) ** sharp
        out[..., :3] += lobe[..., None] * np.array([6.0, 5.0, 3.5])
        for k in range(6):
            c = u[k * 6:k * 6 + 3] * 2 - 1; c /= np.linalg.norm(c)
            col = u[k * 6 + 3:k * 6 + 6] * 2.0
            out[..., :3] += (np.clip((d * c).sum(-1), 0, 1) ** (sharp * 0.5))[..., None] * col
        out[..., 3] = 1.0
        return out

    levels = int(np.log2(env_size)) + 1
    chain = [sky(_cube_dirs(env_size), 64.0)]
    for _l in range(1, levels):
        p = chain[-1]
        s2 = p.shape[1] // 2
        chain.append(p.reshape(6, s2, 2, s2, 2, 4).mean(axis=(2, 4)))
    env_chain = np.concatenate([c.astype(np.float32).reshape(-1) for c in chain])
    irr = sky(_cube_dirs(irr_size), 2.0).astype(np.float32) * np.float32(0.35)
    ao = None
    if with_ao:
        ao = (0.3 + 0.7 * _value_noise(width, height, 32, STREAM_SURFACE, 1 << 25, seed)).astype(np.float32)
    return IblSet(irradiance=np.ascontiguousarray(irr), env_chain=np.ascontiguousarray(env_chain), env_size=env_size, env_levels=levels,
                  brdf_lut=np.ascontiguousarray(lut, np.float32), ao=ao)


@dataclass
class ShadowSet:
    lights_matrices: np.ndarray                 # float32[4,16]
    maps: list = field(default_factory=list)    # 4 numpy arrays: [0] float32[S,S,4], [1..3] float16[S,S]
    size: int = 0


def make_shadow_set(cam: Camera, size: int, seed: int = SEED) -> ShadowSet:
    """lightsMatrices via the native S9 path; maps from an analytic occluder field (64 discs), no rasteriser.
    Cascade 0: RGBA32F (e^{40z}, e^{80z}, -e^{-40z}, e^{-80z}) (ShadowCaster.shader:71-75); cascades 1-3: R16F z."""
    light_world = host.transform_matrix([0, 0, 0, 0], directional_rotation(), [1, 1, 1, 1])
    light_view = host.mat4_inverse(light_world)
    lm = host.csm_matrices(light_view, cam.world, cam.aspect, cam.fov, cam.z_near, cam.z_far)
    maps = []
    for k in range(4):
        u = uniforms(STREAM_SHADOW, 64 * 4, k * 1024, seed).reshape(64, 4).astype(np.float64)
        ys, xs = np.meshgrid((np.arange(size) + 0.5) / size, (np.arange(size) + 0.5) / size, indexing="ij")
        zf = np.full((size, size), 0.25)
        for d in range(64):
            cx, cy, rad, h = u[d, 0], u[d, 1], 0.03 + 0.12 * u[d, 2], 0.3 + 0.55 * u[d, 3]
            inside = (xs - cx) ** 2 + (ys - cy) ** 2 < rad * rad
            zf = np.where(inside, np.maximum(zf, h), zf)
        if k == 0:
            z32 = zf.astype(np.float32)
            e = np.exp(np.float32(40.0) * z32).astype(np.float32)
            n = (-np.exp(np.float32(-40.0) * z32)).astype(np.float32)
            maps.append(np.ascontiguousarray(np.stack([e, e * e, n, n * n], -1).astype(np.float32)))
        else:
            maps.append(np.ascontiguousarray(zf.astype(np.float16)))
    return ShadowSet(lights_matrices=lm, maps=maps, size=size)


@dataclass
class EntitySet:
    transforms: np.ndarray     # float32[N, 12]  (position4, rotation xyzw, scale4)
    parent: np.ndarray         # uint32[N], 0xFFFFFFFF = root, level-sorted
    level_offsets: np.ndarray  # uint32[L+1]
    local_aabb: np.ndarray     # float32[N, 6]


def make_entities(count: int, seed: int = SEED, editor_world: bool = True) -> EntitySet:
    """TRS with position U[-8000,8000]^3, unit quaternion, scale U[0.5,4]; local AABB centre 0, extents U[0.5,50]^3;
    hierarchy depth <= 3, level-sorted, 70 % roots.  With editor_world the first 4 roots are the Editor.world objects
    (Content/Editor.world:4,46,75,104: Camera at (0,150,0), Sponza, Box, Light)."""
    n0 = max(1, int(round(count * 0.7)))
    n1 = int(round(count * 0.2))
    n2 = count - n0 - n1
    if n2 < 0:
        n1 += n2; n2 = 0
    levels = [n0, n1, n2]
    offs = np.cumsum([0] + levels).astype(np.uint32)
    u = uniforms(STREAM_ENTITIES, count * 16, 0, seed).reshape(count, 16).astype(np.float64)
    trs = np.zeros((count, 12), np.float32)
    trs[:, 0:3] = (-8000 + 16000 * u[:, 0:3]).astype(np.float32)
    trs[:, 3] = 1.0
    q = u[:, 3:7] * 2 - 1
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    trs[:, 4:8] = q.astype(np.float32)
    s = (0.5 + 3.5 * u[:, 7:10]).astype(np.float32)
    trs[:, 8:11] = s
    trs[:, 11] = 1.0
    ext = (0.5 + 49.5 * u[:, 10:13]).astype(np.float32)
    aabb = np.concatenate([-ext, ext], 1).astype(np.float32)
    parent = np.full(count, 0xFFFFFFFF, np.uint32)
    pu = u[:, 13]
    for lvl in (1, 2):
        lo, hi = int(offs[lvl]), int(offs[lvl + 1])
        plo, phi = int(offs[lvl - 1]), int(offs[lvl])
        if hi > lo:
            parent[lo:hi] = (plo + np.floor(pu[lo:hi] * (phi - plo))).astype(np.uint32)
            trs[lo:hi, 0:3] *= np.float32(0.01)  # children live near their parent
    if editor_world and count >= 4:
        trs[0] = [0, 150, 0, 1, 0, 0, 0, 1, 1, 1, 1, 1]
        for i in (1, 2):
            trs[i] = [0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1]
        trs[3] = [0, 0, 0, 0, *directional_rotation(), 1, 1, 1, 1]
    level_offsets = offs if n2 > 0 else (offs[:3] if n1 > 0 else offs[:2])
    return EntitySet(transforms=np.ascontiguousarray(trs), parent=parent, level_offsets=np.ascontiguousarray(level_offsets.astype(np.uint32)),
                     local_aabb=np.ascontiguousarray(aabb))


# ---------------------------------------------------------------------------------------------------------------
# The BASELINE.json configurations (C1..C5).  radius_scale ("s") is FROZEN here; the realised list
# statistics of each configuration are recorded in DESIGN.md and in the golden fixtures.
# ---------------------------------------------------------------------------------------------------------------
CONFIGS = {
    "C2": dict(width=1920, height=1080, lights=LightSetConfig(count=4096, spot_fraction=0.0, radius_scale=3.5)),
    "C3": dict(width=3840, height=2160, lights=LightSetConfig(count=65536, spot_fraction=0.25, radius_scale=0.85, cluster_lights=7200, cluster_count=12)),
    "C4": dict(width=3840, height=2160, lights=LightSetConfig(count=65536, spot_fraction=0.25, radius_scale=0.85, cluster_lights=7200, cluster_count=12,
                                                              directional_first=True), shadow_size=4096),
    "C5": dict(width=7680, height=4320, lights=LightSetConfig(count=1048576, spot_fraction=0.25, radius_scale=0.2, cluster_lights=7200, cluster_count=12),
               entities=1048576),
    # small fixtures used by the CPU/GPU parity suites
    "tiny": dict(width=128, height=96, lights=LightSetConfig(count=512, spot_fraction=0.25, radius_scale=6.0, cluster_lights=300, cluster_count=1, cluster_spread=1.5, cluster_radius=(0.5, 1.2))),
    "tiny_csm": dict(width=128, height=96, lights=LightSetConfig(count=512, spot_fraction=0.25, radius_scale=6.0, cluster_lights=300, cluster_count=1,
                                                                 cluster_spread=1.5, cluster_radius=(0.5, 1.2), directional_first=True), shadow_size=64),
}


@dataclass
class Frame:
    name: str
    cam: Camera
    depth: np.ndarray
    lights: np.ndarray
    surface: np.ndarray | None = None
    shadows: ShadowSet | None = None


def make_frame(name: str, with_surface: bool = True, seed: int = SEED, **override) -> Frame:
    cfg = dict(CONFIGS[name]); cfg.update(override)
    cam = make_camera(cfg["width"], cfg["height"])
    depth = make_linear_depth(cam.width, cam.height, seed)
    lights = make_lights(cam, depth, cfg["lights"], seed)
    surface = make_surface(cam, depth, seed) if with_surface else None
    shadows = make_shadow_set(cam, cfg["shadow_size"], seed) if cfg.get("shadow_size") else None
    return Frame(name=name, cam=cam, depth=depth, lights=lights, surface=surface, shadows=shadows)


@dataclass
class InstanceSet:
    instances: np.ndarray   # host.INSTANCE_DTYPE [count]
    batches: np.ndarray     # uint32 [numBatches, 5]: indexCount, instanceCount, firstIndex, vertexOffset, firstInstance


def make_instance_set(count: int, num_batches: int, seed: int = SEED, first_instance: int = 0, spread: float = 3000.0) -> InstanceSet:
    """The per-instance SSBO + indirect buffer of one RHIRecordDrawCallGPUCulling call (RHI/Batch.hpp:146-159): `count` instances
    in `num_batches` contiguous batches of ragged size (a few empty, a few long), in front of `first_instance` untouched records.
    model = translate(U[-spread, spread]^3) * scale(U[0.5, 4]); sphere centre U[-5,5]^3, radius U[1,61]; materialInstance = the
    record's original index (so a moved record can be recognised)."""
    total = first_instance + count
    u = uniforms(STREAM_ENTITIES, total * 8, 1 << 26, seed).reshape(total, 8)
    inst = np.zeros(total, host.INSTANCE_DTYPE)
    sc = (np.float32(0.5) + np.float32(3.5) * u[:, 0]).astype(np.float32)
    model = np.zeros((total, 16), np.float32)
    model[:, 0] = sc; model[:, 5] = sc; model[:, 10] = sc; model[:, 15] = 1.0
    model[:, 12:15] = ((u[:, 1:4] * 2 - 1) * np.float32(spread)).astype(np.float32)
    model[:, 14] -= np.float32(0.5 * spread)  # most of the cloud in front of the camera (it looks down -z)
    inst["model"] = model
    inst["sphereBounds"][:, :3] = (u[:, 4:7] - np.float32(0.5)) * np.float32(10)
    inst["sphereBounds"][:, 3] = np.float32(1) + np.float32(60) * u[:, 7]
    inst["materialInstance"] = np.arange(total, dtype=np.uint32)
    inst["isCulled"] = 7
    # ragged batch sizes: cut points from the stream, every 16th batch empty, every 64th takes a large share
    w = uniforms(STREAM_ENTITIES, num_batches, (1 << 26) + (1 << 25), seed).astype(np.float64) + 0.05
    w[::16] = 0.0
    w[5::64] *= 40.0
    if w.sum() == 0.0:
        w[:] = 1.0
    sizes = np.floor(w / w.sum() * count).astype(np.int64)
    sizes[np.argmax(w)] += count - int(sizes.sum())
    first = first_instance + np.concatenate([[0], np.cumsum(sizes)[:-1]])
    batches = np.zeros((num_batches, 5), np.uint32)
    batches[:, 0] = 36 + 3 * (np.arange(num_batches) % 100)
    batches[:, 1] = sizes
    batches[:, 2] = 1000 * np.arange(num_batches)
    batches[:, 3] = np.arange(num_batches)
    batches[:, 4] = first
    return InstanceSet(instances=inst, batches=batches)


def unit_cube_mesh():
    """positions float32[8, 3] of the cube [-1, 1]^3 and its 12 outward-facing triangles uint32[12, 3] (the casters' stand-in mesh: every entity
    is drawn as its local bounding box)"""
    p = np.float32([[x, y, z] for z in (-1, 1) for y in (-1, 1) for x in (-1, 1)])
    quads = [(0, 2, 3, 1), (4, 5, 7, 6), (0, 1, 5, 4), (2, 6, 7, 3), (0, 4, 6, 2), (1, 3, 7, 5)]
    tris = np.uint32([t for a, b, c, d in quads for t in ((a, b, c), (a, c, d))])
    return p, tris


def caster_models(world: np.ndarray, local_aabb: np.ndarray) -> np.ndarray:
    """PerInstanceData.model of the box casters: world matrix x translate(box centre) x scale(box half extents), float32[N, 16] column-major
    (float64 product rounded once: these are inputs)"""
    w = np.asarray(world, np.float64).reshape(-1, 4, 4).transpose(0, 2, 1)          # -> row-major 4x4
    box = np.asarray(local_aabb, np.float64).reshape(-1, 6)
    centre, half = 0.5 * (box[:, :3] + box[:, 3:]), 0.5 * (box[:, 3:] - box[:, :3])
    local = np.zeros((len(box), 4, 4))
    local[:, 0, 0] = half[:, 0]; local[:, 1, 1] = half[:, 1]; local[:, 2, 2] = half[:, 2]; local[:, 3, 3] = 1.0
    local[:, :3, 3] = centre
    return np.ascontiguousarray((w @ local).transpose(0, 2, 1).reshape(-1, 16).astype(np.float32))


# ---- `.world` text (Content/Editor.world): what WorldPrefab::Serialize writes (WorldPrefabImporter.cpp:18-32, PrefabImporter.cpp:17-50) ----
def _yaml_seq(values, indent: int) -> str:
    pad = " " * indent
    return "".join(f"{pad}- {float(np.float32(v))!r}\n" if not float(np.float32(v)).is_integer() else f"{pad}- {int(v)}\n" for v in values)


def make_world_text(name: str, prefabs: list) -> str:
    """prefabs: a list of prefabs, each a list of game objects {name, position[4], rotation[4] (x, y, z, w), scale[4], parent (index inside the
    prefab or None), components: [{typename, properties: {key: scalar | sequence | dict}}]}.  Numbers are written with enough digits to read
    back as the same float32 (the reference writes %.9g-style floats)."""
    out = [f"name: {name}\n", "prefabs:\n"]
    for objects in prefabs:
        comps = []
        out.append("  - gameObjects:\n")
        for k, go in enumerate(objects):
            out.append(f"      - name: {go['name']}\n")
            for key in ("position", "rotation", "scale"):
                out.append(f"        {key}:\n" + _yaml_seq(go[key], 10))
            parent = go.get("parent")
            out.append(f"        parentIndex: {4294967295 if parent is None else parent}\n")
            out.append(f"        instanceId: {1000 + k}\n")
            out.append("        components:\n")
            for c in go.get("components", []):
                out.append(f"          - {len(comps)}\n")
                comps.append(c)
            if not go.get("components"):
                out[-1] = "        components: []\n"
        out.append("    components:\n" if comps else "    components: []\n")
        for c in comps:
            out.append(f"      - typename: {c['typename']}\n        overrideProperties:\n")
            for key, val in c.get("properties", {}).items():
                if isinstance(val, dict):
                    out.append(f"          {key}:\n" + "".join(f"            {k2}: {v2}\n" for k2, v2 in val.items()))
                elif isinstance(val, (list, tuple, np.ndarray)):
                    out.append(f"          {key}:\n" + _yaml_seq(val, 12))
                else:
                    out.append(f"          {key}: {val}\n")
            out.append("          fileId: NullFileId\n")
    return "".join(out)


def make_triangle_soup(count: int, seed: int = SEED):
    """A deterministic triangle soup around an eye at the origin looking down -Z (positions float32 [3 * count, 3], indices uint32 [count, 3]): sizes over
    three decades, a part of it across the near plane z = -0.1 and behind the eye, some vertices exactly on the plane / at w = 0 -- the rasteriser's clip cases."""
    u = uniforms(STREAM_ENTITIES, count * 16, 7, seed).reshape(count, 16).astype(np.float64)
    centre = np.stack([6 * u[:, 0] - 3, 6 * u[:, 1] - 3, 9.5 * u[:, 2] - 8.0], axis=1)[:, None, :]
    extent = (10.0 ** (2.7 * u[:, 3] - 2.0))[:, None, None]
    verts = (centre + (2 * u[:, 4:13].reshape(count, 3, 3) - 1) * extent).astype(np.float32)
    verts[::41, :, 2] = np.float32(-0.1)
    verts[7::53, 0, 2] = np.float32(0.0)
    return np.ascontiguousarray(verts.reshape(-1, 3)), np.arange(3 * count, dtype=np.uint32).reshape(count, 3)


def perspective_reversed_z(width: int, height: int, near: float = 0.1, f: float = 1.0) -> np.ndarray:
    """column-major float32[16]: reversed Z, infinite far plane, looking down -Z (w = -z, ndc z = near / -z)"""
    m = np.zeros(16, np.float32)
    m[0], m[5], m[2 * 4 + 3], m[3 * 4 + 2] = f * height / width, f, -1.0, near
    return m

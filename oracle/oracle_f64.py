"""Independent float64 restatement of K2 (PBR shade), K3 (cascaded-shadow factor) and K4 (ECS transform / bounds / frustum sweep).

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED by the reference (it has no golden vectors): this file exists to harden the fp32 C oracle
(oracle/sailor_oracle.c), not to replace it.  It was written from the reference's own text only --

  K2   Content/Shaders/Standard.shader:259-341 (CalculateLighting), :377-439 (main), Lighting.glsl:39-76 (NdfGGX, GeometrySchlick*, FresnelSchlick)
  K3   Lighting.glsl:168-197 (ManualPCF), :200-216 (SelectCascade), :218-240 (Linstep / ReduceLightBleed / Chebyshev), :242-284 (the two lookups)
  K4   Runtime/Math/Transform.cpp:39-42, Runtime/ECS/TransformECS.cpp:144-212, Runtime/Math/Bounds.cpp:245-260,479-492, Bounds.h:119-130

-- without consulting the C file, in float64, vectorised over pixels / entities, with library `sqrt`, `exp`, `power` and true divisions:
no evaluation-order games, no fused operations, no fast reciprocals.  What it is for (tests/test_oracle_cpu.py):

  * the fp32 oracle must agree with it to 1e-4 relative wherever the expression is well conditioned; the pixels where it does not are
    listed by cause (NdfGGX's cancelling denominator at the specular peak of smooth surfaces, the edge of a light's radius window or
    spot cone, a PCF compare or a cascade boundary that flips) and bounded in number;
  * K4's fp32 matrices / boxes must agree with it to fp32 rounding and the visibility bits wherever no plane distance is within
    rounding of zero.

The only things shared with the rest of the repository are data layouts (the 112-byte light record, the 232-byte frame UBO, the three
surface planes) and the texture sampling convention that SURVEY.md 8(d) fixes for the synthetic shadow maps (texel centres at
(i + 0.5) / size, bilinear, clamp to edge).
"""
from __future__ import annotations

import numpy as np

TILE = 16
CASCADE_LEVELS = (0.05, 0.1, 0.333333, 0.5)  # Constants.glsl:24
POISSON = np.array([
    (-0.94201624, -0.39906216), (0.94558609, -0.76890725), (-0.094184101, -0.92938870), (0.34495938, 0.29387760),
    (-0.91588581, 0.45771432), (-0.81544232, -0.87912464), (-0.38277543, 0.27676845), (0.97484398, 0.75648379),
    (0.44323325, -0.97511554), (0.53742981, -0.47373420), (-0.26496911, -0.41893023), (0.79197514, 0.19090188),
    (-0.24188840, 0.99706507), (-0.81409955, 0.91437590), (0.19984126, 0.78641367), (0.14383161, -0.14100790)], np.float64)  # Lighting.glsl:176-185
PI = 3.14159265359  # Math.glsl:1
EPSILON = 0.00001   # Standard.shader:258
LIGHT_DTYPE = np.dtype({"names": ["type", "shadowType", "worldPosition", "direction", "intensity", "attenuation", "cutOff", "bounds"],
                        "formats": ["<u4", "<u4", ("<f4", 3), ("<f4", 3), ("<f4", 3), ("<f4", 3), ("<f4", 2), ("<f4", 3)],
                        "offsets": [0, 4, 16, 32, 48, 64, 80, 96], "itemsize": 112})  # Lighting.glsl:4-15


def frame_fields(frame_bytes) -> dict:
    """UboFrameData (RHI/Types.h:751-761): column-major mat4s -> numpy matrices M with M @ v = the GLSL product."""
    b = np.frombuffer(bytes(frame_bytes), np.uint8)
    f = b[:208].view(np.float32).astype(np.float64)
    col = lambda o: f[o:o + 16].reshape(4, 4).T
    return {"view": col(0), "projection": col(16), "invProjection": col(32), "cameraPosition": f[48:51],
            "viewportSize": b[208:216].view(np.int32).astype(np.int64), "cameraZNearZFar": b[216:224].view(np.float32).astype(np.float64)}


def _normalize(v):
    return v / np.sqrt((v * v).sum(-1, keepdims=True))


def _texture_bilinear(tex: np.ndarray, u, v):
    """texture(sampler2D, uv) of a single-level image, linear filter, clamp to edge; tex[H, W] or [H, W, C]; u, v arrays."""
    Hh, Ww = tex.shape[0], tex.shape[1]
    t = tex.astype(np.float64)
    x = u * Ww - 0.5
    y = v * Hh - 0.5
    x0 = np.floor(x); y0 = np.floor(y)
    ax = x - x0; ay = y - y0
    xi0 = np.clip(x0.astype(np.int64), 0, Ww - 1); xi1 = np.clip(x0.astype(np.int64) + 1, 0, Ww - 1)
    yi0 = np.clip(y0.astype(np.int64), 0, Hh - 1); yi1 = np.clip(y0.astype(np.int64) + 1, 0, Hh - 1)
    if t.ndim == 3:
        ax = ax[..., None]; ay = ay[..., None]
    top = t[yi0, xi0] * (1.0 - ax) + t[yi0, xi1] * ax
    bot = t[yi1, xi0] * (1.0 - ax) + t[yi1, xi1] * ax
    return top * (1.0 - ay) + bot * ay


def select_cascade(view, world_pos, z_far):
    """Lighting.glsl:200-216"""
    p = world_pos @ view[:, :3].T + view[:, 3]
    depth = np.abs(p[..., 2] / p[..., 3])
    layer = np.full(depth.shape, 4, np.int64)
    for i in (3, 2, 1, 0):
        layer = np.where(depth < z_far * CASCADE_LEVELS[i], i, layer)
    return layer


def _chebyshev(m0, m1, current, min_variance, linstep):
    d = current - m0
    variance = np.maximum(min_variance, m1 - m0 * m0)
    with np.errstate(divide="ignore", invalid="ignore"):
        pmax = variance / (variance + d * d)
        red = np.clip((pmax - linstep) / (1.0 - linstep), 0.0, 1.0)
    return np.where(d < 0, 1.0, red)


def shadow_pcf(tex, frag_light, bias):
    """Lighting.glsl:242-261 + :168-197 (the 17th fetch at :255 is dead code)"""
    proj = frag_light[..., :3] / frag_light[..., 3:4]
    proj = proj * 0.5 + 0.5
    px, py, pz = proj[..., 0], 1.0 - proj[..., 1], proj[..., 2]
    outside = (px > 1.0) | (py > 1.0) | (px < 0.0) | (py < 0.0) | (pz < 0.5)
    texel = 1.0 / np.array([tex.shape[1], tex.shape[0]], np.float64)
    shadow = np.zeros(px.shape)
    for i in range(16):
        off = POISSON[i] * 2.0 * texel
        pcf_depth = _texture_bilinear(tex, px + off[0], py + off[1]) * 0.5 + 0.5
        shadow += np.where(pz + bias > pcf_depth, 1.0, 0.0)
    return np.where(outside, 1.0, shadow / 16.0)


def shadow_evsm(tex, frag_light, bias, cascade):
    """Lighting.glsl:263-284"""
    proj = frag_light[..., :3] / frag_light[..., 3:4]
    px, py, pz = proj[..., 0] * 0.5 + 0.5, 1.0 - (proj[..., 1] * 0.5 + 0.5), proj[..., 2]
    outside = (px > 1.0) | (py > 1.0) | (px < 0.0) | (py < 0.0) | (pz < 0.0)
    s = _texture_bilinear(tex, px, py)
    current = np.exp(40.0 * (pz + 0.003 * bias * np.power(0.5, cascade)))
    neg_current = -np.exp(-40.0 * (pz + 0.0001 * bias))
    pos_value = _chebyshev(s[..., 0], s[..., 1], current, 0.01, 0.0)
    neg_value = _chebyshev(s[..., 2], s[..., 3], neg_current, 0.0, 0.0) * np.where(cascade > 2, 0.0, 1.0)
    return np.where(outside, 1.0, np.clip(1.0 - np.maximum(pos_value, neg_value), 0.0, 1.0))


def directional_shadow(fr, light_direction, shadow_type, normal, world_pos, lights_matrices, maps):
    """Standard.shader:266-283.  lights_matrices: float[4, 16] column-major; maps: 4 images (cascade 0 RGBA, 1..3 single channel) or None entries."""
    cascade = np.minimum(select_cascade(fr["view"], world_pos, fr["cameraZNearZFar"][1]), 3)
    ndl = (normal * light_direction).sum(-1)
    out = np.ones(cascade.shape)
    wp1 = np.concatenate([world_pos, np.ones(world_pos.shape[:-1] + (1,))], -1)
    for c in range(4):
        sel = cascade == c
        if not sel.any() or maps[c] is None:
            continue
        M = np.asarray(lights_matrices[c], np.float64).reshape(4, 4).T
        frag = wp1[sel] @ M.T
        if shadow_type == 2 and c == 0:
            bias = (1.0 - ndl[sel]) * (1 + c)
            out[sel] = shadow_evsm(np.asarray(maps[c]), frag, bias, np.full(frag.shape[0], c))
        else:
            bias = np.maximum(0.000075 * (1.0 - ndl[sel]), 0.000005)
            tex = np.asarray(maps[c])
            out[sel] = shadow_pcf(tex if tex.ndim == 2 else tex[..., 0], frag, bias)
    return out


def calculate_lighting(fr, L, albedo, metallic, roughness, F0, Lo, cos_lo, normal, world_pos, csm):
    """Standard.shader:259-341 for ONE light over an array of pixels -> float64[..., 3]"""
    ltype = int(L["type"])
    pos = L["worldPosition"].astype(np.float64); direction = L["direction"].astype(np.float64)
    att = L["attenuation"].astype(np.float64); cut = L["cutOff"].astype(np.float64)
    falloff = np.ones(world_pos.shape[:-1]); shadow = np.ones(world_pos.shape[:-1])
    with np.errstate(divide="ignore", invalid="ignore"):
        if ltype == 0:
            if csm is not None:
                shadow = directional_shadow(fr, direction, int(L["shadowType"]), normal, world_pos, csm[0], csm[1])
        elif ltype == 1:
            distance = np.sqrt(((pos - world_pos) ** 2).sum(-1))
            attenuation = 1.0 / (att[0] + att[1] * distance + att[2] * (distance * distance))
            falloff = attenuation * (1.0 - np.power(np.clip(distance / float(L["bounds"][0]), 0.0, 1.0), 2.0))
        elif ltype == 2:
            light_dir = _normalize(pos - world_pos)
            epsilon = cut[0] - cut[1]
            theta = (light_dir * _normalize(-direction)).sum(-1)
            distance = np.sqrt(((pos - world_pos) ** 2).sum(-1))
            attenuation = 1.0 / (att[0] + att[1] * distance + att[2] * (distance * distance))
            falloff = attenuation * np.clip((theta - cut[1]) / epsilon, 0.0, 1.0)
            falloff = np.where(theta < cut[1], 0.0, falloff)
        Li = -direction
        Lh = _normalize(Li + Lo)
        cos_li = np.maximum(0.0, (normal * Li).sum(-1))
        cos_lh = np.maximum(0.0, (normal * Lh).sum(-1))
        F = F0 + (1.0 - F0) * np.power(1.0 - np.maximum(0.0, (Lh * Lo).sum(-1)), 5.0)[..., None]
        alpha = roughness * roughness
        alpha_sq = alpha * alpha
        denom = (cos_lh * cos_lh) * (alpha_sq - 1.0) + 1.0
        D = alpha_sq / (PI * denom * denom)
        r = roughness + 1.0
        k = (r * r) / 8.0
        G = (cos_li / (cos_li * (1.0 - k) + k)) * (cos_lo / (cos_lo * (1.0 - k) + k))
        kd = (1.0 - F) * (1.0 - metallic)[..., None]  # mix(1 - F, 0, metallic)
        diffuse = kd * albedo
        specular = (F * (D * G)[..., None]) / np.maximum(EPSILON, 4.0 * cos_li * cos_lo)[..., None]
        return shadow[..., None] * ((diffuse + specular) * L["intensity"].astype(np.float64) * cos_li[..., None]) * falloff[..., None]


def shade(frame_bytes, W: int, H: int, surface: np.ndarray, lights: np.ndarray, grid: np.ndarray, indices: np.ndarray, csm=None, rows=None,
          want_conditioning: bool = False):
    """Standard.shader:377-439 over the synthetic surface: surface float32[3, H, W, 4] = (worldPos, albedo.a) (normal, roughness)
    (albedo.rgb, metallic); grid uint32[T, 2], indices uint32[...] the canonical cull output; csm = (lightsMatrices[4, 16], [4 maps]) or None.
    -> radiance float64[H, W, 4] (ambient term 0).  With want_conditioning also returns, per pixel, the smallest NdfGGX denominator met."""
    fr = frame_fields(frame_bytes)
    lights = np.asarray(lights).view(LIGHT_DTYPE).reshape(-1) if np.asarray(lights).dtype != LIGHT_DTYPE else np.asarray(lights)
    s = surface.astype(np.float64)
    out = np.zeros((H, W, 4))
    min_denom = np.full((H, W), np.inf)
    vw, vh = int(fr["viewportSize"][0]), int(fr["viewportSize"][1])
    tiles_x = vw // TILE + min(1, vw % TILE)
    r0, r1 = (0, H) if rows is None else rows
    ys, xs = np.mgrid[r0:r1, 0:W]
    # gl_FragCoord = pixel centre; screenUv = (x, viewportSize.y - y); tileId = ivec2(screenUv) / 16
    tile_x = np.floor(xs + 0.5).astype(np.int64) // TILE
    tile_y = np.floor(vh - (ys + 0.5)).astype(np.int64) // TILE
    tile_index = tile_y * tiles_x + tile_x
    for t in np.unique(tile_index):
        m = tile_index == t
        py, px = ys[m], xs[m]
        world_pos = s[0, py, px, :3]; albedo_a = s[0, py, px, 3]
        normal = s[1, py, px, :3]; roughness = s[1, py, px, 3]
        albedo = s[2, py, px, :3]; metallic = s[2, py, px, 3]
        view_dir = _normalize(world_pos - fr["cameraPosition"])
        cos_lo = np.maximum(0.0, (normal * -view_dir).sum(-1))
        F0 = 0.04 * (1.0 - metallic)[..., None] + albedo * metallic[..., None]  # mix(Fdielectric, albedo, metallic)
        acc = np.zeros(world_pos.shape)
        offset, num = int(grid[t, 0]), int(grid[t, 1])
        for i in range(num):
            index = int(indices[offset + i])
            if index == 0xFFFFFFFF:
                break
            acc += calculate_lighting(fr, lights[index], albedo, metallic, roughness, F0, -view_dir, cos_lo, normal, world_pos, csm)
            if want_conditioning:
                Lh = _normalize(-lights[index]["direction"].astype(np.float64) - view_dir)
                cl = np.maximum(0.0, (normal * Lh).sum(-1))
                a2 = (roughness * roughness) ** 2
                min_denom[py, px] = np.minimum(min_denom[py, px], cl * cl * (a2 - 1.0) + 1.0)
        out[py, px, :3] = acc
        out[py, px, 3] = albedo_a
    return (out, min_denom) if want_conditioning else out


# ---- K4 ---------------------------------------------------------------------------------------------------------------------------------
def _quat_to_mat(q):
    """glm::toMat4(quat(x, y, z, w)) -- the rotation matrix of a quaternion (not necessarily unit: glm does not normalise)"""
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.zeros(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - 2 * (y * y + z * z); R[..., 0, 1] = 2 * (x * y - w * z); R[..., 0, 2] = 2 * (x * z + w * y)
    R[..., 1, 0] = 2 * (x * y + w * z); R[..., 1, 1] = 1 - 2 * (x * x + z * z); R[..., 1, 2] = 2 * (y * z - w * x)
    R[..., 2, 0] = 2 * (x * z - w * y); R[..., 2, 1] = 2 * (y * z + w * x); R[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def transform_matrix(trs):
    """Math/Transform.cpp:39-42: translate(position) * toMat4(rotation) * scale(scale); trs float[..., 12] = position.xyzw, rotation.xyzw, scale.xyzw"""
    trs = np.asarray(trs, np.float64)
    M = np.zeros(trs.shape[:-1] + (4, 4))
    M[..., :3, :3] = _quat_to_mat(trs[..., 4:8]) * trs[..., None, 8:11]
    M[..., :3, 3] = trs[..., 0:3]
    M[..., 3, 3] = 1.0
    return M


def ecs_sweep(trs, parent, local_aabb, planes):
    """TransformECS::Tick full sweep + CalculateMatrices (ECS/TransformECS.cpp:144-212), AABB::Apply (Math/Bounds.cpp:479-492, with its
    FLT_MIN seed of the maximum), Frustum::OverlapsAABB (Math/Bounds.cpp:245-260).  parent: uint32, 0xFFFFFFFF = root, parents before children.
    -> world float64[n, 4, 4] (M @ v), world_aabb float64[n, 6] (min, max), visible bool[n], margin float64[n] (smallest |plane distance|)"""
    rel = transform_matrix(trs)
    n = len(parent)
    world = np.zeros_like(rel)
    for i in range(n):
        p = int(parent[i])
        world[i] = rel[i] if p == 0xFFFFFFFF else world[p] @ rel[i]
    la = np.asarray(local_aabb, np.float64)
    mn, mx = la[:, :3], la[:, 3:]
    corners = np.stack([np.stack([np.where(b & 1, mx[:, 0], mn[:, 0]), np.where(b & 2, mx[:, 1], mn[:, 1]), np.where(b & 4, mx[:, 2], mn[:, 2])], -1)
                        for b in range(8)], 1)                                     # [n, 8, 3]
    wc = np.einsum("nij,nkj->nki", world[:, :3, :3], corners) + world[:, None, :3, 3]
    flt_min = float(np.finfo(np.float32).tiny)                                      # numeric_limits<float>::min(): the reference's seed of m_max
    wmin = wc.min(1)
    wmax = np.maximum(wc.max(1), flt_min)
    pl = np.asarray(planes, np.float64).reshape(6, 4)
    # Bounds.cpp:245-260: for each plane  sum_i max(min_i n_i, max_i n_i) + d > 0
    r = np.maximum(wmin[:, None, :] * pl[None, :, :3], wmax[:, None, :] * pl[None, :, :3]).sum(-1) + pl[None, :, 3]
    return world, np.concatenate([wmin, wmax], 1), (r > 0).all(1), np.abs(r).min(1)

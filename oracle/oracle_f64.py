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


# =====================================================================================================================================
# Round 3: the "next" rows a second time -- EVSM blur, irradiance cube, pre-filtered environment cube -- written from the shader
# text (Lighting.glsl:83-127, ComputeIrradianceMap.shader, ComputeEnvMap_IBL.shader, Math.glsl:285-293, Lighting.glsl:27-48) and, for what the
# shaders leave to the sampler, from the Vulkan specification's cube-map rules (major-axis face selection table, (sc / |ma| + 1) / 2, ties: z, then
# y, then x), bilinear inside the face with clamp-to-edge, linear between the two nearest mips, lod clamped to the chain.  float64, vectorised over
# samples; only the data layout (level-major / face / row / texel RGBA32F) is shared with the C oracle.  The depth rasteriser's second restatement
# is the exact-integer / exact-rational one in tests/test_oracle_cpu.py (round 2).
# =====================================================================================================================================
EVSM_BLUR_WEIGHTS = np.array([  # Lighting.glsl:87-99
    [0.5, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0],
    [0.281088, 0.218912, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0],
    [0.197159, 0.176426, 0.126415, 0, 0, 0, 0, 0, 0, 0, 0, 0],
    [0.152068, 0.142855, 0.118431, 0.0866459, 0, 0, 0, 0, 0, 0, 0, 0],
    [0.123827, 0.118971, 0.105518, 0.0863909, 0.0652929, 0, 0, 0, 0, 0, 0, 0],
    [0.104454, 0.101593, 0.0934699, 0.0813492, 0.0669741, 0.0521595, 0, 0, 0, 0, 0, 0],
    [0.0903332, 0.0885083, 0.083252, 0.0751759, 0.0651684, 0.0542336, 0.0433285, 0, 0, 0, 0, 0],
    [0.07958, 0.0783462, 0.0747585, 0.0691403, 0.061977, 0.0538465, 0.0453433, 0.0370081, 0, 0, 0, 0],
    [0.0711171, 0.0702445, 0.0676904, 0.0636383, 0.0583697, 0.0522315, 0.0455989, 0.0388376, 0.0322721, 0, 0, 0],
    [0.0642825, 0.0636429, 0.0617619, 0.0587498, 0.0547779, 0.0500633, 0.0448484, 0.0393811, 0.0338957, 0.0285966, 0, 0],
    [0.0586472, 0.0581645, 0.0567402, 0.0544433, 0.0513831, 0.0476999, 0.0435548, 0.039118, 0.0345572, 0.0300277, 0.0256641, 0],
    [0.0539209, 0.0535478, 0.0524437, 0.050654, 0.0482506, 0.0453272, 0.0419936, 0.0383686, 0.034573, 0.0307232, 0.0269255, 0.0232718]], np.float64)


def evsm_blur_pass(image: np.ndarray, radius_umbra: int, radius_penumbra: int, vertical: bool) -> np.ndarray:
    """GaussianBlur_Evsm (Lighting.glsl:83-127) as one pass of Blur.shader {EVSM, HORIZONTAL | VERTICAL}: image [H, W, 4], radius = (umbra, penumbra);
    the taps uv +- i texelSize of a fragment at a texel centre are texel centres: the texel itself, clamp-to-edge."""
    img = np.asarray(image, np.float64)
    H, W, _ = img.shape
    step_count = 12
    blur_radius = min(max(radius_umbra, radius_penumbra), step_count)
    r1, r2 = min(radius_umbra, step_count), min(radius_penumbra, step_count)
    out = np.zeros_like(img)
    axis, n = (0, H) if vertical else (1, W)
    idx = np.arange(n)
    for i in range(blur_radius):
        plus, minus = np.clip(idx + i, 0, n - 1), np.clip(idx - i, 0, n - 1)
        both = np.take(img, plus, axis=axis) + np.take(img, minus, axis=axis)
        if i < radius_umbra:
            out[..., 2:4] += both[..., 2:4] * EVSM_BLUR_WEIGHTS[r1 - 1][i]
        if i < radius_penumbra:
            out[..., 0:2] += both[..., 0:2] * EVSM_BLUR_WEIGHTS[r2 - 1][i]
    return out


def _cube_levels(chain: np.ndarray, size0: int, levels: int):
    """the flat RGBA32F mip chain as a list of [6, size, size, 4] float64 arrays"""
    out, o = [], 0
    flat = np.asarray(chain, np.float64).reshape(-1)
    for l in range(levels):
        sz = max(size0 >> l, 1)
        out.append(flat[o:o + 6 * sz * sz * 4].reshape(6, sz, sz, 4))
        o += 6 * sz * sz * 4
    return out


def _cube_face_st(d):
    """Vulkan spec "Cube Map Face Selection": (face, s, t) of direction vectors d [n, 3]; ties go to z, then y, then x"""
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    ax, ay, az = np.abs(x), np.abs(y), np.abs(z)
    use_z = (az >= ax) & (az >= ay)
    use_y = ~use_z & (ay >= ax)
    use_x = ~use_z & ~use_y
    face = np.where(use_z, np.where(z < 0, 5, 4), np.where(use_y, np.where(y < 0, 3, 2), np.where(x < 0, 1, 0)))
    sc = np.where(use_z, np.where(z < 0, -x, x), np.where(use_y, x, np.where(x < 0, z, -z)))
    tc = np.where(use_z, -y, np.where(use_y, np.where(y < 0, -z, z), -y))
    ma = np.where(use_z, az, np.where(use_y, ay, ax))
    with np.errstate(invalid="ignore", divide="ignore"):
        return face, 0.5 * (sc / ma + 1.0), 0.5 * (tc / ma + 1.0)


def _cube_bilinear(level: np.ndarray, face, s, t):
    size = level.shape[1]
    x, y = s * size - 0.5, t * size - 0.5
    fx, fy = np.floor(x), np.floor(y)
    ax, ay = (x - fx)[:, None], (y - fy)[:, None]
    x0, y0 = fx.astype(np.int64), fy.astype(np.int64)
    x1, y1 = np.clip(x0 + 1, 0, size - 1), np.clip(y0 + 1, 0, size - 1)
    x0, y0 = np.clip(x0, 0, size - 1), np.clip(y0, 0, size - 1)
    top = level[face, y0, x0] * (1.0 - ax) + level[face, y0, x1] * ax
    bot = level[face, y1, x0] * (1.0 - ax) + level[face, y1, x1] * ax
    return top * (1.0 - ay) + bot * ay


def cube_texture_lod(levels_list, d, lod):
    """textureLod(samplerCube, d, lod): rgba [n, 4]"""
    face, s, t = _cube_face_st(d)
    lod = np.clip(lod, 0.0, len(levels_list) - 1.0)
    l0 = np.floor(lod).astype(np.int64)
    l1 = np.minimum(l0 + 1, len(levels_list) - 1)
    f = (lod - l0)[:, None]
    out = np.zeros((len(d), 4))
    for l in range(len(levels_list)):
        m0, m1 = l0 == l, l1 == l
        if m0.any():
            out[m0] += _cube_bilinear(levels_list[l], face[m0], s[m0], t[m0]) * (1.0 - f[m0])
        if m1.any():
            out[m1] += _cube_bilinear(levels_list[l], face[m1], s[m1], t[m1]) * f[m1]
    return out


def _radical_inverse_vdc(i):
    """Math.glsl:285-293: the 32-bit reversal of i times 2^-32"""
    b = np.asarray(i, np.uint64) & 0xFFFFFFFF
    b = ((b << 16) | (b >> 16)) & 0xFFFFFFFF
    b = (((b & 0x55555555) << 1) | ((b & 0xAAAAAAAA) >> 1)) & 0xFFFFFFFF
    b = (((b & 0x33333333) << 2) | ((b & 0xCCCCCCCC) >> 2)) & 0xFFFFFFFF
    b = (((b & 0x0F0F0F0F) << 4) | ((b & 0xF0F0F0F0) >> 4)) & 0xFFFFFFFF
    b = (((b & 0x00FF00FF) << 8) | ((b & 0xFF00FF00) >> 8)) & 0xFFFFFFFF
    return b.astype(np.float64) * 2.3283064365386963e-10


def _sampling_vector(gx, gy, face, size):
    """GetSamplingVector (ComputeIrradianceMap.shader / ComputeEnvMap_IBL.shader): the direction of output texel (gx, gy) of `face`"""
    st = np.array([gx / size, gy / size])
    uv = 2.0 * np.array([st[0], 1.0 - st[1]]) - 1.0
    ret = [(1.0, uv[1], -uv[0]), (-1.0, uv[1], uv[0]), (uv[0], 1.0, -uv[1]), (uv[0], -1.0, uv[1]), (uv[0], uv[1], 1.0), (-uv[0], uv[1], -1.0)][face]
    ret = np.array(ret, np.float64)
    return ret / np.linalg.norm(ret)


def _basis(N):
    """ComputeBasisVectors: T = cross(N, up) unless degenerate (dot(T, T) < Epsilon), then cross(N, x); S = normalize(cross(N, T))"""
    T = np.cross(N, (0.0, 1.0, 0.0))
    if not (np.dot(T, T) >= 0.00001):   # step(Epsilon, dot(T, T)) == 0
        T = np.cross(N, (1.0, 0.0, 0.0))
    T = T / np.linalg.norm(T)
    S = np.cross(N, T)
    return S / np.linalg.norm(S), T


TWO_PI = 2.0 * PI  # Constants.glsl (TwoPI = 2 * PI with Math.glsl's PI)


def compute_irradiance_map(env_chain: np.ndarray, env_size: int, env_levels: int, size: int, num_samples: int = 64 * 1024) -> np.ndarray:
    """ComputeIrradianceMap.shader main(): [6, size, size, 4] float64 (alpha 1)"""
    levels_list = _cube_levels(env_chain, env_size, env_levels)
    i = np.arange(num_samples)
    u1, u2 = i / float(num_samples), _radical_inverse_vdc(i)          # SampleHammersley: (i * InvNumSamples, RadicalInverse_VdC(i))
    u1p = np.sqrt(np.maximum(0.0, 1.0 - u1 * u1))                      # SampleHemisphere(u1, u2)
    hemi = np.stack([np.cos(TWO_PI * u2) * u1p, np.sin(TWO_PI * u2) * u1p, u1], 1)
    out = np.zeros((6, size, size, 4))
    for face in range(6):
        for gy in range(size):
            for gx in range(size):
                N = _sampling_vector(gx, gy, face, size)
                S, T = _basis(N)
                Li = hemi[:, 0:1] * S + hemi[:, 1:2] * T + hemi[:, 2:3] * N
                cos_theta = np.maximum(0.0, Li @ N)
                rgb = cube_texture_lod(levels_list, Li, np.zeros(num_samples))[:, :3]
                out[face, gy, gx, :3] = (2.0 * rgb * cos_theta[:, None]).sum(0) / num_samples
                out[face, gy, gx, 3] = 1.0
    return out


def prefilter_env_level(raw_chain: np.ndarray, size0: int, levels: int, level: int, roughness: float, num_samples: int = 1024) -> np.ndarray:
    """ComputeEnvMap_IBL.shader main() for one output level: [6, s, s, 4] float64 with s = size0 >> level"""
    levels_list = _cube_levels(raw_chain, size0, levels)
    s_out = max(size0 >> level, 1)
    wt = 4.0 * PI / (6.0 * size0 * size0)
    i = np.arange(num_samples)
    u1, u2 = i / float(num_samples), _radical_inverse_vdc(i)
    alpha = roughness * roughness                                       # SampleGGX (Lighting.glsl:27-37)
    cos_t = np.sqrt((1.0 - u2) / (1.0 + (alpha * alpha - 1.0) * u2))
    sin_t = np.sqrt(1.0 - cos_t * cos_t)
    phi = TWO_PI * u1
    ggx = np.stack([sin_t * np.cos(phi), sin_t * np.sin(phi), cos_t], 1)
    out = np.zeros((6, s_out, s_out, 4))
    for face in range(6):
        for gy in range(s_out):
            for gx in range(s_out):
                N = _sampling_vector(gx, gy, face, s_out)
                S, T = _basis(N)
                Lh = ggx[:, 0:1] * S + ggx[:, 1:2] * T + ggx[:, 2:3] * N
                Li = 2.0 * (Lh @ N)[:, None] * Lh - N                   # Lo = N
                cos_li = Li @ N
                m = cos_li > 0.0
                cos_lh = np.maximum(Lh[m] @ N, 0.0)
                alpha_sq = alpha * alpha                                # NdfGGX (Lighting.glsl:41-48)
                denom = cos_lh * cos_lh * (alpha_sq - 1.0) + 1.0
                with np.errstate(divide="ignore", invalid="ignore"):
                    pdf = alpha_sq / (PI * denom * denom) * 0.25
                    ws = 1.0 / (num_samples * pdf)
                    mip = np.maximum(0.5 * np.log2(ws / wt) + 1.0, 0.0)
                rgb = cube_texture_lod(levels_list, Li[m], mip)[:, :3]
                weight = cos_li[m].sum()
                out[face, gy, gx, :3] = (rgb * cos_li[m][:, None]).sum(0) / weight
                out[face, gy, gx, 3] = 1.0
    return out

"""Second, independent CPU restatement of the tile light cull (K1) in NumPy float32.

TEST INFRASTRUCTURE ONLY (see sailor_oracle.c).  PARITY UNPINNED by the reference (no golden vectors exist); this file and
the C restatement were written separately from the shader text and must agree BIT FOR BIT on `lightsGrid`/`culledLights`
before either is trusted (SURVEY.md 8c "Oracle design") -- tests/test_oracle_cpu.py enforces that.

Follows Content/Shaders/ComputeLightCulling.shader:49-240, Math.glsl:116-173,224-239 under the canonical sequential
semantics of SURVEY.md Appendix A.  Every arithmetic step is an explicit float32 numpy operation in the order the GLSL
writes it (numpy ufuncs never fuse a multiply with an add).  The nearest-128 selection uses the *sort* formulation
(stable descending sort, keep the last 128, emit reversed), not the rank formulation the C file uses.
"""
from __future__ import annotations

import numpy as np

F = np.float32
TILE, CAND, KEEP = 16, 196, 128


def _frame_fields(frame_bytes: bytes):
    b = np.frombuffer(bytes(frame_bytes), np.uint8)
    f = b[:208].view(np.float32)
    view = f[0:16].reshape(4, 4)           # view[c][r]
    inv_proj = f[32:48].reshape(4, 4)
    vp = b[208:216].view(np.int32)
    return view, inv_proj, int(vp[0]), int(vp[1])


def _mul_glsl(M, x, y, z, w):
    """GLSL mat4 * vec4: ((c0*x + c1*y) + c2*z) + c3*w, per row."""
    return [((M[0][i] * x + M[1][i] * y) + M[2][i] * z) + M[3][i] * w for i in range(4)]


def _screen_to_view(inv_proj, sx, sy, sz, sw, vp_w, vp_h, F=F):
    tx = F(sx) / F(vp_w)
    ty = F(sy) / F(vp_h)
    cx = tx * F(2.0) - F(1.0)
    cy = ty * F(2.0) - F(1.0)
    v = _mul_glsl(inv_proj, cx, cy, F(sz), F(sw))
    w = v[3]
    return np.array([v[0] / w, v[1] / w, (v[2] / w) * F(-1.0)], F)


def _plane(p1, p2, F=F):
    """ComputePlane(eye = 0, p1, p2) -> unit normal (plane.w = dot(n, 0) = +-0 never changes a comparison)."""
    v0 = p1 - F(0.0)
    v2 = p2 - F(0.0)
    c = np.array([v0[1] * v2[2] - v2[1] * v0[2], v0[2] * v2[0] - v2[2] * v0[0], v0[0] * v2[1] - v2[0] * v0[1]], F)
    ln = np.sqrt((c[0] * c[0] + c[1] * c[1]) + c[2] * c[2])
    return c / ln


def tile_frustum(inv_proj, tx, ty, vp_w, vp_h, F=F):
    x0, y0, x1, y1 = F(tx * TILE), F(ty * TILE), F((tx + 1) * TILE), F((ty + 1) * TILE)
    vs0 = _screen_to_view(inv_proj, x0, y0, -1.0, 1.0, vp_w, vp_h, F)
    vs1 = _screen_to_view(inv_proj, x1, y0, -1.0, 1.0, vp_w, vp_h, F)
    vs2 = _screen_to_view(inv_proj, x0, y1, -1.0, 1.0, vp_w, vp_h, F)
    vs3 = _screen_to_view(inv_proj, x1, y1, -1.0, 1.0, vp_w, vp_h, F)
    vs4 = _screen_to_view(inv_proj, (x0 + x1) * F(0.5), (y0 + y1) * F(0.5), (F(-1.0) + F(-1.0)) * F(0.5), (F(1.0) + F(1.0)) * F(0.5), vp_w, vp_h, F)
    planes = [_plane(vs2, vs0, F), _plane(vs1, vs3, F), _plane(vs0, vs1, F), _plane(vs3, vs2, F)]
    return planes, vs4[0], vs4[1]


def overlap_table(frame_bytes, W: int, H: int, lights: np.ndarray, depth: np.ndarray, dtype=np.float32):
    """The per-(tile, light) decisions of Math.glsl:224-239 SphereFrustumOverlaps (+ the directional rule of :153-162) evaluated in `dtype`,
    and how close each decision is to flipping: `slack[t, j]` = the smallest relative distance of any of the six comparisons from its
    boundary.  With dtype = float64 this is the "truth" the float32 restatements are held against (tests/test_oracle_cpu.py): the two may
    only disagree where float64 itself says the sphere touches a plane or a depth bound to within rounding."""
    T = dtype
    view32, inv32, vp_w, vp_h = _frame_fields(frame_bytes)
    view, inv_proj = view32.astype(T), inv32.astype(T)
    Tx, Ty = (W - 1) // TILE + 1, (H - 1) // TILE + 1
    ltype = lights["type"].astype(np.uint32)
    radius = lights["bounds"][:, 0].astype(T)
    wp = lights["worldPosition"].astype(T)
    p = _mul_glsl(view, wp[:, 0], wp[:, 1], wp[:, 2], T(1.0))
    px, py, pz = p[0] / p[3], p[1] / p[3], (p[2] / p[3]) * T(-1.0)
    depth_bits = np.ascontiguousarray(depth, np.float32).view(np.uint32)
    ok = np.zeros((Ty * Tx, len(lights)), bool)
    slack = np.full((Ty * Tx, len(lights)), np.inf)
    lx = np.arange(TILE)
    scale = np.abs(px) + np.abs(py) + np.abs(pz) + np.abs(radius)
    with np.errstate(invalid="ignore", divide="ignore"):
        for ty in range(Ty):
            rows = np.clip(H - 1 - (TILE * ty + lx), 0, H - 1)
            for tx in range(Tx):
                cols = np.minimum(TILE * tx + lx, W - 1)
                bits = depth_bits[np.ix_(rows, cols)]
                z_far, z_near = T(bits.max().view(np.float32)), T(bits.min().view(np.float32))
                diff = z_far - z_near
                z_far, z_near = z_far - diff, z_near + diff
                planes, _, _ = tile_frustum(inv_proj, tx, ty, vp_w, vp_h, T)
                o = ~((pz - radius > z_near) | (pz + radius < z_far))
                s = np.minimum(np.abs((pz - radius) - z_near), np.abs((pz + radius) - z_far)) / (scale + np.abs(z_near) + np.abs(z_far))
                for pl in planes:
                    d = (pl[0] * px + pl[1] * py) + pl[2] * pz
                    o &= ~(d < -radius)
                    s = np.minimum(s, np.abs(d + radius) / scale)
                o |= ltype == 0
                s = np.where(ltype == 0, np.inf, s)
                ok[ty * Tx + tx] = o
                slack[ty * Tx + tx] = s
    return ok, slack


def light_cull(frame_bytes, W: int, H: int, lights: np.ndarray, depth: np.ndarray, tile_rows=None):
    view, inv_proj, vp_w, vp_h = _frame_fields(frame_bytes)
    Tx, Ty = (W - 1) // TILE + 1, (H - 1) // TILE + 1
    r0, r1 = (0, Ty) if tile_rows is None else tile_rows
    n = len(lights)
    ltype = lights["type"].astype(np.uint32)
    radius = lights["bounds"][:, 0].astype(F)
    wp = lights["worldPosition"].astype(F)
    if n:
        p = _mul_glsl(view, wp[:, 0], wp[:, 1], wp[:, 2], F(1.0))
        w = p[3]
        px, py, pz = p[0] / w, p[1] / w, (p[2] / w) * F(-1.0)
    depth_bits = np.ascontiguousarray(depth, F).view(np.uint32)

    grid = np.zeros(((r1 - r0) * Tx, 2), np.uint32)
    out = [np.zeros(1, np.uint32)]
    running = 0
    lx = np.arange(TILE)
    for ty in range(r0, r1):
        rows = np.clip(H - 1 - (TILE * ty + lx), 0, H - 1)
        for tx in range(Tx):
            cols = np.minimum(TILE * tx + lx, W - 1)
            bits = depth_bits[np.ix_(rows, cols)]
            max_d = bits.max().view(F)   # atomicMax / atomicMin on float bits (:125-126)
            min_d = bits.min().view(F)
            z_far, z_near = max_d, min_d
            diff = z_far - z_near
            z_far = z_far - diff
            z_near = z_near + diff
            if n == 0:
                cand = np.zeros(0, np.int64); impact = np.zeros(0, F)
            else:
                planes, cx, cy = tile_frustum(inv_proj, tx, ty, vp_w, vp_h)
                ok = ~((pz - radius > z_near) | (pz + radius < z_far))
                for pl in planes:
                    d = (pl[0] * px + pl[1] * py) + pl[2] * pz
                    ok &= ~(d < -radius)
                ok |= ltype == 0
                cand = np.nonzero(ok)[0][:CAND]             # ascending light index, first 196
                cz = (z_far + z_near) * F(0.5)
                dx, dy, dz = px[cand] - cx, py[cand] - cy, pz[cand] - cz
                impact = np.sqrt((dx * dx + dy * dy) + dz * dz).astype(F)
                impact[ltype[cand] == 0] = F(0.0)
            k = len(cand)
            if k > KEEP and np.isnan(impact).any():
                # no total order (sky tile: depth +inf -> NaN centre): the shader's partial bubble sort literally,
                # ComputeLightCulling.shader:198-225 -- a compare next to a NaN is false, so NaNs never move
                cand = cand.copy(); imp = impact.copy()
                left = KEEP
                for i in range(k - 1):
                    for j in range(k - i - 1):
                        if imp[j] < imp[j + 1]:
                            imp[j], imp[j + 1] = imp[j + 1], imp[j]
                            cand[j], cand[j + 1] = cand[j + 1], cand[j]
                    left -= 1
                    if left == 0:
                        break
            elif k > KEEP:
                order = np.argsort(-impact, kind="stable")   # descending impact, ties keep original order
                cand = cand[order]
            num = min(k, KEEP)
            lst = cand[::-1][:num].astype(np.uint32)         # emit reversed: the last `num` slots
            t = (ty - r0) * Tx + tx
            grid[t] = (running + 1, num)
            out.append(lst)
            running += num
    indices = np.concatenate(out)
    indices[0] = running
    return grid, indices

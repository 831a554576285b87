"""Known answers worked out by hand from the shader's formulas (Standard.shader:286-340): independent of the oracle's two restatements and of the
HIP path.  `shade` = the implementation under test: oracle.shade by default, the HIP path in tests/test_shade_gpu.py."""
import numpy as np

from oracle import oracle
from sailor_amd import host, synth


def one_light_frame(light_type, roughness, metallic, albedo, dist, radius, attenuation, intensity, cut_off=None, off_axis=None, shade=None, shadow_type=0, csm=None):
    """The tiny frame with ONE pixel set up for a closed form: its normal points at the camera (n = Lo), the light hangs `dist` along the normal and
    its `direction` field is -n (the shader's Li is -light.direction for every type), so cosLi = cosLh = cosLo = 1 and Lh = n."""
    f = synth.make_frame("tiny")
    W, H = f.cam.width, f.cam.height
    surface = f.surface.copy()
    py, px = H // 3, W // 3
    wp = surface[0, py, px, :3].astype(np.float64)
    cam = np.asarray(f.cam.world, np.float64)[12:15]
    n = (cam - wp) / np.linalg.norm(cam - wp)
    surface[1, py, px, :3] = n.astype(np.float32)
    surface[1, py, px, 3] = roughness
    surface[2, py, px] = np.float32(list(albedo) + [metallic])
    lights = np.zeros(1, host.LIGHT_DTYPE)
    lights["type"] = light_type
    lights["shadowType"] = shadow_type
    pos = wp + n * dist
    axis = n
    if off_axis is not None:   # a spot light whose axis is tilted by `off_axis` radians away from the direction to the surface point
        t = np.cross(n, [0.0, 0.0, 1.0]); t /= np.linalg.norm(t)
        axis = np.cos(off_axis) * n + np.sin(off_axis) * t
    lights["worldPosition"] = pos.astype(np.float32)
    lights["direction"] = (-axis).astype(np.float32)
    lights["intensity"] = np.float32(intensity)
    lights["attenuation"] = np.float32(attenuation)
    lights["bounds"] = np.float32([radius] * 3)
    if cut_off is not None:
        lights["cutOff"] = np.float32(cut_off)
    Tx, Ty = oracle.num_tiles(W, H)
    grid = np.zeros((Tx * Ty, 2), np.uint32); grid[:, 0] = 1 + np.arange(Tx * Ty); grid[:, 1] = 1   # every tile: the list [0]
    idx = np.zeros(1 + Tx * Ty, np.uint32); idx[0] = Tx * Ty
    out = (shade or oracle.shade)(f.cam.frame, W, H, surface, lights, grid, idx, *([csm] if csm is not None else []))
    # what float32 storage made of the set-up (the closed form is evaluated on the stored values)
    n32 = surface[1, py, px, :3].astype(np.float64)
    d32 = np.linalg.norm(lights["worldPosition"][0].astype(np.float64) - wp)
    return out[py, px].astype(np.float64), n32, d32


def point_light_at_normal_incidence(roughness, metallic, shade=None):
    """(radiance of the pixel, its closed form): specular = F0 / (4 pi a^2), kd = (1 - F0)(1 - metallic), falloff = (1 - (d / r)^2) / (a.x + a.y d + a.z d^2)."""
    albedo, intensity, att, d, r = (0.8, 0.5, 0.25), (3.0, 2.0, 5.0), (1.0, 0.022, 0.0019), 40.0, 100.0
    got, _n32, d32 = one_light_frame(host.LIGHT_POINT, roughness, metallic, albedo, d, r, att, intensity, shade=shade)
    a = float(np.float32(roughness)) ** 2
    F0 = 0.04 * (1 - metallic) + np.array(albedo) * metallic
    spec = F0 / (4 * np.pi * a * a)
    kd = (1 - F0) * (1 - metallic)
    falloff = (1 - min(max(d32 / r, 0), 1) ** 2) / (att[0] + att[1] * d32 + att[2] * d32 * d32)
    return got, (kd * np.array(albedo) + spec) * np.array(intensity) * falloff

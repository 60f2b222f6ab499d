"""Host-side emitter preparation that feeds next-event estimation.

Restates, in float32 numpy arithmetic, the scene-load step the reference runs
once per scene on the CPU:
  collect_emitters              librender/lights.cpp:14-73
  estimate_normalized_radiance  librender/lights.cpp:166-199
  trim_dim_emitters             librender/lights.cpp:201-217
  equalize_emitter_bins         librender/lights.cpp:220-349
  update_light_sampling         librender/lights.cpp:75-90
  halton2                       util/compute_util.h:19-33
In a drop-in build the reference's own librender provides these (the adapter in
INTEGRATION.md calls them); this module exists so the synthetic scenes can be
prepared without the reference.
"""
import bisect
import math

import numpy as np

f32 = np.float32


def halton2(index: int) -> np.float32:
    index &= 0xFFFFFFFF
    index = ((index << 16) | (index >> 16)) & 0xFFFFFFFF
    index = (((index & 0x00FF00FF) << 8) | ((index & 0xFF00FF00) >> 8)) & 0xFFFFFFFF
    index = (((index & 0x0F0F0F0F) << 4) | ((index & 0xF0F0F0F0) >> 4)) & 0xFFFFFFFF
    index = (((index & 0x33333333) << 2) | ((index & 0xCCCCCCCC) >> 2)) & 0xFFFFFFFF
    index = (((index & 0x55555555) << 1) | ((index & 0xAAAAAAAA) >> 1)) & 0xFFFFFFFF
    u = np.array([0x3F800000 | (index >> 9)], dtype=np.uint32)
    return f32(u.view(np.float32)[0] - f32(1.0))


def luminance(c):
    return f32(f32(f32(0.2126) * c[0] + f32(0.7152) * c[1]) + f32(0.0722) * c[2])


def _dot(a, b):
    return f32(f32(a[0] * b[0] + a[1] * b[1]) + a[2] * b[2])


def _normalize(v):
    return (v * f32(f32(1.0) / np.sqrt(_dot(v, v)))).astype(f32)


def _cross(a, b):
    return np.array([a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1]], dtype=f32)


def _fma(a, b, c):
    # single-rounded a*b+c for float32 operands: exact in float64, rounded once
    return f32(np.float64(a) * np.float64(b) + np.float64(c))


def triangle_solid_angle(v0, v1, v2):
    """lights.cpp:127-163 (host version with a true atan)."""
    householder_sign = f32(-1.0) if v0[0] > 0 else f32(1.0)
    s = f32(f32(1.0) / f32(abs(v0[0]) + f32(1.0)))
    hy, hz = f32(v0[1] * s), f32(v0[2] * s)
    dot_0_1 = _dot(v0, v1)
    dot_0_2 = _dot(v1, v2)
    dot_1_2 = _dot(v0, v2)
    dh0 = _fma(-householder_sign, v1[0], dot_0_1)
    dh2 = _fma(-householder_sign, v2[0], dot_1_2)
    m00, m01 = _fma(-dh0, hy, v1[1]), _fma(-dh0, hz, v1[2])
    m10, m11 = _fma(-dh2, hy, v2[1]), _fma(-dh2, hz, v2[2])
    simplex_volume = f32(abs(f32(m00 * m11 - m10 * m01)))
    tangent = f32(simplex_volume / f32(f32(f32(1.0) + dot_0_1) + f32(dot_0_2 + dot_1_2)))
    offset = f32(math.pi) if tangent < 0 else f32(0.0)
    return f32(f32(2.0) * f32(f32(np.arctan(tangent)) + offset))


def collect_emitters(scene):
    """lights.cpp:14-73. Returns (n, 4, 3) float32: v0, v1, v2, radiance."""
    from . import scenes as S
    emitters = []
    nonemissive = set()
    for inst in scene.instances:
        pm_id = inst.pmesh
        if pm_id in nonemissive:
            continue
        pm = scene.pmeshes[pm_id]
        mesh = scene.meshes[pm.mesh]
        nxt = []
        tri_base = 0
        for j in range(mesh.num_geometries):
            g = scene.geometries[mesh.first_geometry + j]
            offs = int(pm.material_offsets[j])
            per_tri = pm.tri_material_ids is not None
            if not per_tri:
                mat = scene.materials[offs]
                if not (mat.emission_intensity > 0.0):
                    tri_base += g.num_tris
                    continue
            pos = S.dequantize_positions(g.qpos, g.scaling, g.offset).reshape(-1, 3, 3)
            for t in range(g.num_tris):
                if per_tri:
                    mat = scene.materials[offs + int(pm.tri_material_ids[tri_base + t])]
                    if not (mat.emission_intensity > 0.0):
                        continue
                rad = (f32(mat.emission_intensity) * np.array(mat.base_color[:], dtype=f32)).astype(f32)
                M = inst.transform  # 3x4 row-major object->world
                # float32 affine transform, association of glm mat4*vec4: (m0*x + m1*y) + (m2*z + m3*1)
                v = []
                for k in range(3):
                    p = pos[t, k]
                    w = np.zeros(3, dtype=f32)
                    for r in range(3):
                        w[r] = f32(f32(M[r, 0] * p[0] + M[r, 1] * p[1]) + f32(M[r, 2] * p[2] + M[r, 3]))
                    v.append(w)
                nxt.append(np.stack([v[0], v[1], v[2], rad]))
            tri_base += g.num_tris
        if nxt:
            emitters = nxt + emitters  # emitters.insert(emitters.begin(), ...)
        else:
            nonemissive.add(pm_id)
    if not emitters:
        return np.zeros((0, 4, 3), dtype=f32)
    return np.stack(emitters).astype(f32)


def estimate_normalized_radiance(emitters, min_perceived_receiver_dist):
    """lights.cpp:166-199. Note: the reference divides by M_2_PI (= 2/pi), reproduced."""
    out = np.zeros(len(emitters), dtype=f32)
    d = f32(min_perceived_receiver_dist)
    for i, e in enumerate(emitters):
        v0, v1, v2, rad = e
        c = _cross((v1 - v0).astype(f32), (v2 - v0).astype(f32))
        with np.errstate(all="ignore"):
            n = _normalize(c)
            ln = np.sqrt(_dot(n, n))
        if not (abs(f32(ln - f32(1.0))) < f32(0.05)):
            out[i] = 0.0
            continue
        cen = ((v0 + v1 + v2).astype(f32) / f32(3.0)).astype(f32)
        o = (n * d).astype(f32)
        sa = triangle_solid_angle(_normalize((v0 - cen - o).astype(f32)), _normalize((v1 - cen - o).astype(f32)),
                                  _normalize((v2 - cen - o).astype(f32)))
        # `luminance(..) * (solid_angle / M_2_PI)`: M_2_PI is a double, so quotient AND product are doubles, rounded once on the store (lights.cpp:195)
        out[i] = f32(np.float64(luminance(rad)) * (np.float64(sa) / 0.63661977236758134308))
    return out


def trim_dim_emitters(emitters, radiances, min_radiance):
    keep = radiances >= f32(min_radiance)
    return emitters[keep], radiances[keep]


def equalize_emitter_bins(emitters, radiances, bin_size):
    """lights.cpp:220-349. Returns (emitters', radiances')."""
    n = len(radiances)
    if bin_size <= 1 or n == 0:
        return emitters, radiances
    original_bin_count = (n + (bin_size - 1)) // bin_size
    average_weight = f32(0.0)
    for r in radiances:
        average_weight = f32(average_weight + r)
    average_weight = f32(average_weight / f32(n))
    bins = []  # [radiance, source_idx, split_count]
    for i in range(n):
        w = radiances[i]
        with np.errstate(all="ignore"):
            q = f32(w / average_weight)
        q = min(float(q), float(original_bin_count)) if q == q else float(original_bin_count)
        clones = max(int(np.uint32(int(q)) if q >= 0 else 0), 1)
        for _ in range(clones):
            bins.append([f32(radiances[i] / f32(clones)), i, clones])

    def reshuffle(bins):
        count = len(bins)
        out = [None] * count
        for index in range(count):
            src = int(np.uint32(int(f32(halton2(index) * f32(count)))))
            while True:
                if src >= count:
                    src = 0
                if bins[src][1] == -1:
                    src += 1
                else:
                    break
            out[index] = list(bins[src])
            bins[src][1] = -1
        return out

    def measure_equality(bins):
        mn, mx = f32(2.0e32), f32(0.0)
        i = 0
        while i < len(bins):
            tot = f32(0.0)
            j = 0
            while j < bin_size and i < len(bins):
                tot = f32(tot + bins[i][0])
                i += 1
                j += 1
            mn = min(tot, mn)
            mx = max(tot, mx)
        with np.errstate(all="ignore"):
            return min(f32(mn / mx), f32(1.0))

    bins = reshuffle(bins)
    equality = measure_equality(bins)
    retries = 0
    while equality < f32(0.6) and retries < 2:
        postfix = []
        acc = None
        for k, b in enumerate(bins):
            if k == 0:
                acc = [b[0], b[1], b[2]]
            else:
                acc = [f32(acc[0] + b[0]), b[1], 1]
            postfix.append(list(acc))
        postfix[0][2] = 1
        total = postfix[-1][0]
        for b in postfix:
            b[0] = f32(b[0] / total)
        prev_elements = len(bins)
        prev_bin_count = (prev_elements + (bin_size - 1)) // bin_size
        padded = (prev_bin_count + 1) * bin_size
        keys = [float(b[0]) for b in postfix]
        h = 0
        while len(bins) < padded:
            u = float(halton2(h))
            h += 1
            it = bisect.bisect_right(keys, u)  # upper_bound: first element with u < radiance
            if it == len(postfix):
                it = len(postfix) - 1
            postfix[it][2] += 1
            bins.append([postfix[it][0], it, 0])
        for i in range(prev_elements, padded):
            clone = bins[i]
            original = bins[clone[1]]
            cc = postfix[clone[1]][2]
            if cc > 1:
                original[0] = f32(original[0] / f32(cc))
                original[2] *= cc
                postfix[clone[1]][2] = 1
            clone[0] = original[0]
            clone[1] = original[1]
            clone[2] = original[2]
        bins = reshuffle(bins)
        equality = measure_equality(bins)
        retries += 1
    new_rad = np.array([b[0] for b in bins], dtype=f32)
    new_em = np.stack([emitters[b[1]].copy() for b in bins]).astype(f32)
    for i, b in enumerate(bins):
        new_em[i, 3] = (new_em[i, 3] / f32(b[2])).astype(f32)
    return new_em, new_rad


def update_light_sampling(emitters, min_perceived_receiver_dist=15.0, min_radiance=0.0, bin_size=16):
    """lights.cpp:75-90 starting from an invalidated BinnedLightSampling."""
    if len(emitters) == 0:
        return emitters, np.zeros(0, dtype=f32)
    radiances = estimate_normalized_radiance(emitters, min_perceived_receiver_dist)
    if min_radiance > 0.0:
        emitters, radiances = trim_dim_emitters(emitters, radiances, min_radiance)
    return equalize_emitter_bins(emitters, radiances, bin_size)

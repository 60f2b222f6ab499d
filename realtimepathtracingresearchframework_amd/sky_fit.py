"""Host half of row a15 in Python: the Hosek-Wilkie sky fit + the sun's radiance for a scene state (sun direction, turbidity, ground
albedo) -> abi.SceneParams; the twin of host/sky_fit.hpp (same operations in the same order: doubles through libm, float32 where
RenderVulkan::update_sky_light, vulkan/render_sky.cpp:25-72, uses float).

The model's coefficient tables are READ AT RUN TIME from the data headers an integration points at (RPTR_SKY_DATA / `where`): the
model's own distribution (ArHosekSkyModelData_RGB.h, ArHosekSkyModelData_Spectral.h) or the reference's copies
(rendering/lights/sky_model_arhosek/sky_model_data_{rgb,spectral}.h) plus the CIE 1931 table of rendering/color/color_matching.h.
Nothing of them is stored in this package. References: rendering/lights/sky_model_arhosek/sky_model.cpp:150-348,524-566,608-642,663-822."""
import math
import os
import re

import numpy as np

from . import abi

f32 = np.float32
PI = 3.141592653589793


def parse_c_arrays(path):
    """every `type name[...] = { numbers };` of a C header -> {name: [float, ...]} (comments skipped, Float(x) wrappers accepted)"""
    text = open(path, errors="ignore").read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    out = {}
    for m in re.finditer(r"(\w+)\s*\[[^\]]*\]\s*=\s*\{([^}]*)\}", text):
        body = re.sub(r"\b[A-Za-z_]\w*\s*\(", "(", m.group(2))
        if re.search(r"[A-Za-z_]\w*", re.sub(r"(?<=[\d.])[fF]\b", "", re.sub(r"[eE][-+]?\d+", "", body))):
            continue   # a table of pointers to other arrays
        vals = [float(v.rstrip("fF")) for v in re.findall(r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?[fF]?", body)]
        if vals:
            out[m.group(1)] = vals
    return out


class SkyTables:
    def __init__(self, where=None):
        where = where or os.environ.get("RPTR_SKY_DATA", "")
        dirs = []
        for d in [p for p in where.split(":") if p]:
            for sub in ("", "/lights/sky_model_arhosek", "/color", "/rendering/lights/sky_model_arhosek", "/rendering/color", "/../../color"):
                dirs.append(d + sub)

        def find(*names):
            for d in dirs:
                for n in names:
                    if os.path.isfile(os.path.join(d, n)):
                        return os.path.join(d, n)
            return None
        rgb = find("sky_model_data_rgb.h", "ArHosekSkyModelData_RGB.h")
        if rgb is None:
            raise FileNotFoundError("no sky_model_data_rgb.h / ArHosekSkyModelData_RGB.h under %r (RPTR_SKY_DATA)" % where)
        a = parse_c_arrays(rgb)
        self.rgb = [a["datasetRGB%d" % (c + 1)] for c in range(3)]
        self.rgb_rad = [a["datasetRGBRad%d" % (c + 1)] for c in range(3)]
        assert all(len(v) >= 1080 for v in self.rgb) and all(len(v) >= 120 for v in self.rgb_rad), rgb
        self.note = ""
        spec, cie = find("sky_model_data_spectral.h", "ArHosekSkyModelData_Spectral.h"), find("color_matching.h")
        self.has_sun = spec is not None and cie is not None
        if self.has_sun:
            s = parse_c_arrays(spec)
            wl = [str(320 + 40 * k) for k in range(11)]
            self.spec = [s["dataset" + w] for w in wl]
            self.spec_rad = [s["datasetRad" + w] for w in wl]
            self.solar = [s["solarDataset" + w] for w in wl]
            self.limb = [s["limbDarkeningDataset" + w] for w in wl]
            self.cie = [f32(v) for v in parse_c_arrays(cie)["cie1931_tbl"]]
            assert len(self.cie) >= 285 and all(len(v) >= 1800 for v in self.solar)
        else:
            self.note = "sky fitted, sun left dark: spectral data / colour-matching table not found under %r" % where


def _c(fn):
    """libm semantics: a domain error is NaN, an overflow is inf (Python raises instead)"""
    def g(*a):
        try:
            return fn(*a)
        except ValueError:
            return float("nan")
        except OverflowError:
            return float("inf")
    return g


_pow, _exp, _sqrt = _c(math.pow), _c(math.exp), _c(math.sqrt)


def _bezier(m, i, stride, x):
    p = _pow
    return (p(1.0 - x, 5.0) * m[i] + 5.0 * p(1.0 - x, 4.0) * x * m[i + stride] + 10.0 * p(1.0 - x, 3.0) * p(x, 2.0) * m[i + 2 * stride] +
            10.0 * p(1.0 - x, 2.0) * p(x, 3.0) * m[i + 3 * stride] + 5.0 * (1.0 - x) * p(x, 4.0) * m[i + 4 * stride] + p(x, 5.0) * m[i + 5 * stride])


def cook_configuration(dataset, turbidity, albedo, solar_elevation):
    """sky_model.cpp:150-230"""
    it = int(turbidity)
    rem = turbidity - float(it)
    x = _pow(solar_elevation / (PI / 2.0), (1.0 / 3.0))
    cfg = [0.0] * 9
    base = 9 * 6 * (it - 1)
    for i in range(9):
        cfg[i] = (1.0 - albedo) * (1.0 - rem) * _bezier(dataset, base + i, 9, x)
    base = 9 * 6 * 10 + 9 * 6 * (it - 1)
    for i in range(9):
        cfg[i] += albedo * (1.0 - rem) * _bezier(dataset, base + i, 9, x)
    if it == 10:
        return cfg
    base = 9 * 6 * it
    for i in range(9):
        cfg[i] += (1.0 - albedo) * rem * _bezier(dataset, base + i, 9, x)
    base = 9 * 6 * 10 + 9 * 6 * it
    for i in range(9):
        cfg[i] += albedo * rem * _bezier(dataset, base + i, 9, x)
    return cfg


def cook_radiance_configuration(dataset, turbidity, albedo, solar_elevation):
    """sky_model.cpp:232-292"""
    it = int(turbidity)
    rem = turbidity - float(it)
    x = _pow(solar_elevation / (PI / 2.0), (1.0 / 3.0))
    res = (1.0 - albedo) * (1.0 - rem) * _bezier(dataset, 6 * (it - 1), 1, x)
    res += albedo * (1.0 - rem) * _bezier(dataset, 6 * 10 + 6 * (it - 1), 1, x)
    if it == 10:
        return res
    res += (1.0 - albedo) * rem * _bezier(dataset, 6 * it, 1, x)
    res += albedo * rem * _bezier(dataset, 6 * 10 + 6 * it, 1, x)
    return res


def _radiance_internal(c, theta, gamma):
    """sky_model.cpp:294-307"""
    expM = _exp(c[4] * gamma)
    rayM = math.cos(gamma) * math.cos(gamma)
    mieM = (1.0 + math.cos(gamma) * math.cos(gamma)) / _pow((1.0 + c[8] * c[8] - 2.0 * c[8] * math.cos(gamma)), 1.5)
    zenith = _sqrt(math.cos(theta))
    return (1.0 + c[0] * _exp(c[1] / (math.cos(theta) + 0.01))) * (c[2] + c[3] * expM + c[5] * rayM + c[6] * mieM + c[7] * zenith)


def _spectral_radiance(configs, radiances, theta, gamma, wavelength):
    """sky_model.cpp:524-566"""
    low = int((wavelength - 320.0) / 40.0)
    if low < 0 or low >= 11:
        return 0.0
    interp = math.fmod((wavelength - 320.0) / 40.0, 1.0)
    val_low = _radiance_internal(configs[low], theta, gamma) * radiances[low] * 1.0
    if interp < 1e-6:
        return val_low
    result = (1.0 - interp) * val_low
    if low + 1 < 11:
        result += interp * _radiance_internal(configs[low + 1], theta, gamma) * radiances[low + 1] * 1.0
    return result


def _sr_internal(t, turbidity, wl, elevation):
    """sky_model.cpp:663-692"""
    pieces, order = 45, 4
    v = _pow(2.0 * elevation / PI, 1.0 / 3.0) * pieces
    pos = int(v) if v == v and abs(v) < 2e9 else -2147483648   # (int) of NaN on x86-64
    pos = min(pos, 44)
    break_x = _pow((float(pos) / float(pieces)), 3.0) * (PI * 0.5)
    at = order * pieces * turbidity + order * (pos + 1) - 1
    res, x, x_exp = 0.0, elevation - break_x, 1.0
    for _ in range(order):
        res += x_exp * t.solar[wl][at]
        at -= 1
        x_exp *= x
    return res * 1.0


def _solar_radiance_internal2(t, st_turbidity, solar_radius, wavelength, elevation, gamma):
    """sky_model.cpp:694-796"""
    turb_low = int(st_turbidity) - 1
    turb_frac = st_turbidity - float(turb_low + 1)
    if turb_low == 9:
        turb_low, turb_frac = 8, 1.0
    wl_low = int((wavelength - 320.0) / 40.0)
    wl_frac = math.fmod(wavelength, 40.0) / 40.0
    if wl_low == 10:
        wl_low, wl_frac = 9, 1.0
    direct = ((1.0 - turb_frac) * ((1.0 - wl_frac) * _sr_internal(t, turb_low, wl_low, elevation) + wl_frac * _sr_internal(t, turb_low, wl_low + 1, elevation)) +
              turb_frac * ((1.0 - wl_frac) * _sr_internal(t, turb_low + 1, wl_low, elevation) + wl_frac * _sr_internal(t, turb_low + 1, wl_low + 1, elevation)))
    ld = [(1.0 - wl_frac) * t.limb[wl_low][i] + wl_frac * t.limb[wl_low + 1][i] for i in range(6)]
    sol_rad_sin = math.sin(solar_radius)
    ar2 = 1 / (sol_rad_sin * sol_rad_sin)
    singamma = math.sin(gamma)
    sc2 = max(1.0 - ar2 * singamma * singamma, 0.0)
    sc = math.sqrt(sc2)
    dark = ld[0] + ld[1] * sc + ld[2] * _pow(sc, 2.0) + ld[3] * _pow(sc, 3.0) + ld[4] * _pow(sc, 4.0) + ld[5] * _pow(sc, 5.0)
    return direct * dark


def fit_sky(tables, sun_dir, turbidity, albedo, light_count, normal_z_scale=1.0):
    """update_sky_light (vulkan/render_sky.cpp:25-72) -> abi.SceneParams"""
    d = np.asarray(sun_dir, f32)
    inv = f32(1.0) / np.sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2], dtype=f32)
    d = (d * inv).astype(f32)
    al = np.asarray(albedo, f32)
    a_avg = (al[0] * f32(0.3333) + al[1] * f32(0.3333)) + al[2] * f32(0.3333)
    T, A, E = float(f32(turbidity)), float(a_avg), float(d[1])
    sp = abi.SceneParams()
    cfg = [cook_configuration(tables.rgb[c], T, A, E) for c in range(3)]
    for i in range(9):
        sp.sky_params.configs[i][:] = [float(f32(cfg[0][i])), float(f32(cfg[1][i])), float(f32(cfg[2][i])), 0.0]
    sp.sky_params.radiances[:] = [float(f32(cook_radiance_configuration(tables.rgb_rad[c], T, A, E))) for c in range(3)] + [0.0]
    sp.sun_dir[:] = [float(v) for v in d]
    sp.sun_cos_angle = float(_cosf((f32(0.53) * f32(0.01745329251994329576923690768489)) / f32(2.0)))
    sun = [0.0, 0.0, 0.0, 0.0]
    if tables.has_sun:
        solar_radius = (0.51 * (PI / 180.0)) / 2.0
        configs = [cook_configuration(tables.spec[w], T, A, E) for w in range(11)]
        radiances = [cook_radiance_configuration(tables.spec_rad[w], T, A, E) for w in range(11)]
        xyz = [f32(0), f32(0), f32(0)]
        n, last = 0, f32(360.0)
        for i in range(95):
            wavelength = f32(i) * f32(830.0 - 360.0) / f32(94) + f32(360.0)
            if wavelength > f32(720.0):
                break
            w = float(wavelength)
            theta = float(d[1])
            insc = _spectral_radiance(configs, radiances, theta, 0.0, w)
            radiance = f32(_solar_radiance_internal2(tables, T, solar_radius, w, (PI / 2.0) - theta, 0.0) + insc)
            radiance = f32(float(radiance) - insc)
            for k in range(3):
                xyz[k] = f32(xyz[k] + f32(tables.cie[k * 95 + i] * radiance))
            n += 1
            last = wavelength
        scale = f32(last - f32(360.0)) / f32(n)
        xyz = [f32(v * scale) for v in xyz]
        M = [[f32(3.240479), f32(-1.537150), f32(-0.498535)], [f32(-0.969256), f32(1.875991), f32(0.041556)], [f32(0.055648), f32(-0.204043), f32(1.057311)]]
        if d[1] > 0 and all(v >= 0 for v in xyz):
            sun = [float(f32(0.01) * f32(f32(f32(M[r][0] * xyz[0]) + f32(M[r][1] * xyz[1])) + f32(M[r][2] * xyz[2]))) for r in range(3)] + [1.0]
    sun[3] = sun[3] * 0.5 if light_count > 0 else 1.0
    sp.sun_radiance[:] = sun
    sp.normal_z_scale = normal_z_scale
    return sp


def _libm_f32(name):
    """cosf / sinf of the C library: what std::cos(float) is in the C++ host (numpy's float32 kernels may differ from it in the last bit)"""
    import ctypes
    import ctypes.util
    try:
        fn = getattr(ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6"), name)
        fn.restype, fn.argtypes = ctypes.c_float, [ctypes.c_float]
        return lambda x: f32(fn(float(x)))
    except Exception:
        return {"cosf": lambda x: np.cos(f32(x), dtype=f32), "sinf": lambda x: np.sin(f32(x), dtype=f32)}[name]


_cosf, _sinf = _libm_f32("cosf"), _libm_f32("sinf")


def sun_dir_from_height_angle(height_deg, angle_deg):
    """the "Sun" sliders of the reference's scene state (libapp/scene_state.h:79-96)"""
    rad = f32(0.01745329251994329576923690768489)
    ct, st = _cosf(rad * (f32(90.0) - f32(height_deg))), _sinf(rad * (f32(90.0) - f32(height_deg)))
    return np.array([_cosf(rad * f32(angle_deg)) * st, ct, _sinf(rad * f32(angle_deg)) * st], f32)

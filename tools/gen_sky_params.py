"""Generates realtimepathtracingresearchframework_amd/data/sky_params.json (package data: the fitted sky of the built-in scenes) from the reference's own Hosek-Wilkie
fit (rendering/lights/sky_model_arhosek/sky_model.cpp compiled unmodified into
oracle/_ref/libsky_ref.so; driver oracle/ref_sky_driver.cpp restates
vulkan/render_sky.cpp:25-72). Run in the build container only:

    make -C oracle ref && python tools/gen_sky_params.py
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from realtimepathtracingresearchframework_amd.scenes import SKY_CONFIGS  # noqa: E402


class RefSkyOut(C.Structure):
    _fields_ = [("configs", (C.c_float * 4) * 9), ("radiances", C.c_float * 4), ("sun_dir", C.c_float * 3),
                ("sun_cos_angle", C.c_float), ("sun_radiance", C.c_float * 4)]


def main():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libsky_ref.so"))
    lib.ref_update_sky_light.argtypes = [C.POINTER(C.c_float), C.c_float, C.POINTER(C.c_float), C.c_int, C.POINTER(RefSkyOut)]
    entries = {}
    for key, cfg in SKY_CONFIGS.items():
        e = {"input": cfg}
        for tag, lc in (("nolights", 0), ("lights", 1)):
            out = RefSkyOut()
            sd = (C.c_float * 3)(*cfg["sun_dir"])
            al = (C.c_float * 3)(*cfg["albedo"])
            lib.ref_update_sky_light(sd, cfg["turbidity"], al, lc, C.byref(out))
            e["configs"] = [[float(out.configs[i][j]) for j in range(4)] for i in range(9)]
            e["radiances"] = [float(x) for x in out.radiances]
            e["sun_dir"] = [float(x) for x in out.sun_dir]
            e["sun_cos_angle"] = float(out.sun_cos_angle)
            e["sun_radiance_" + tag] = [float(x) for x in out.sun_radiance]
        entries[key] = e
    # evaluation golden vectors for the sun-at-zenith config: the reference's C evaluation
    # (arhosek_tristim_skymodel_radiance) on directions where theta == gamma
    import numpy as np
    lib.ref_sky_radiance.argtypes = [C.POINTER(C.c_float), C.c_float, C.POINTER(C.c_float), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    cfg = SKY_CONFIGS["default"]
    cos_t = np.linspace(0.0, 1.0, 33)
    theta = np.arccos(cos_t)
    out = np.zeros((len(theta), 3))
    lib.ref_sky_radiance((C.c_float * 3)(*cfg["sun_dir"]), cfg["turbidity"], (C.c_float * 3)(*cfg["albedo"]),
                         theta.ctypes.data, theta.ctypes.data, len(theta), out.ctypes.data)
    entries["default"]["eval_cos_theta"] = [float(x) for x in cos_t]
    entries["default"]["eval_rgb_times_100"] = [[float(v) for v in row] for row in out]
    # ... and for EVERY configuration on the directions where the zenith angle equals the angle to the sun (the great circle through the
    # bisector of up and sun, perpendicular to up - sun): there the shader's formula, which feeds the zenith angle into the term the C
    # code feeds the sun angle into (sky_model.glsl:46-48), must agree with the reference's C evaluation
    for key, cfg in SKY_CONFIGS.items():
        sd = np.array(entries[key]["sun_dir"], np.float64)
        up = np.array([0.0, 1.0, 0.0])
        b = up + sd
        nrm = np.cross(up, sd)
        if np.linalg.norm(b) < 1e-6 or np.linalg.norm(nrm) < 1e-6:
            continue  # sun at the zenith (covered above) or at the nadir
        b /= np.linalg.norm(b)
        nrm /= np.linalg.norm(nrm)
        phi = np.linspace(-1.45, 1.45, 25)
        dirs = np.cos(phi)[:, None] * b[None, :] + np.sin(phi)[:, None] * nrm[None, :]
        dirs = dirs[dirs[:, 1] > 0.02]
        theta = np.arccos(np.clip(dirs[:, 1], -1, 1))
        out = np.zeros((len(theta), 3))
        lib.ref_sky_radiance((C.c_float * 3)(*cfg["sun_dir"]), cfg["turbidity"], (C.c_float * 3)(*cfg["albedo"]), theta.ctypes.data, theta.ctypes.data,
                             len(theta), out.ctypes.data)
        entries[key]["eval_equal_angle_dirs"] = [[float(v) for v in row] for row in dirs]
        entries[key]["eval_equal_angle_rgb_times_100"] = [[float(v) for v in row] for row in out]
    doc = {"generator": "tools/gen_sky_params.py", "source": "oracle/_ref/libsky_ref.so <- reference sky_model.cpp + render_sky.cpp:25-72",
           "entries": entries}
    with open(os.path.join(ROOT, "realtimepathtracingresearchframework_amd", "data", "sky_params.json"), "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote", len(entries), "entries")


if __name__ == "__main__":
    main()

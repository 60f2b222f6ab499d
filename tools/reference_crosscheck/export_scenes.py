#!/usr/bin/env python3
"""Exports the BASELINE configurations as files a REAL rptr (the Vulkan reference, on a machine with a Vulkan-RT GPU) loads, next to the
command lines that render them with both programs -- the run that produces "an image from the reference" (north_star: < 1e-3 RMSE), which
this image cannot make (no Vulkan, no GLM / GLFW: SURVEY.md section 8c).

    python tools/reference_crosscheck/export_scenes.py --out crosscheck [--configs c1,c2g,c3] [--small]

Per configuration <name> the directory <out>/<name>/ receives
  <name>.vks + <name>_textures/     the scene in the reference's own format (vks.write_vks; ext/libvkr/src/vkr.c reads it). Material
                                    parameters travel as textures: the reference's host refuses literal base colours / roughness
                                    (vulkan/render_vulkan.cpp:1778-1796 "Material %d is missing a base_color texture") -- a 4 x 4 BC1 block of
                                    one colour each (BaseColor: sRGB, Specular: g = roughness, b = metallic, r = specular), emission and
                                    transmission in the material's .txt (librender/scene.cpp:770-960)
  <name>.ini                        the reference's ImGui-settings configuration (--config): batch spp, path depth, variant, the Sun header
                                    (height / angle / turbidity / Color: libapp/scene_state.h:76-93), light bin size
  commands.sh, compare_images.py    the two command lines (cmdline.cpp:296-474: --validation <prefix> --validation-spp N --pfm --img W H
                                    --eye .. --center .. --up .. --fov ..) and the comparison (bin/rptr_compare = util/compare_exr.cpp)
  manifest.json                     what was exported (triangles, instances, materials as the textures decode them, camera, spp)

c1  Cornell box, 32 triangles, 256 x 256, 1 spp batches up to 64 spp          (BASELINE configs[0])
c2g the 1 M-triangle height field, 1920 x 1080, 4 spp batches, glTF BSDF      (configs[1]'s geometry; its "diffuse-only BSDF" is this
                                                                               build's variant -- the reference ships no such program)
c3  the same + 512 emissive triangles, binned-RIS NEE, 8 spp batches           (configs[2]: the reference's default renderer)
--small: 1/100 of the triangles and 480 x 270 (a quick end-to-end check of the procedure; tests/test_reference_crosscheck.py uses it).

What the numbers of such a run can and cannot say: traversal is the driver's in the reference, texture filtering the sampler's, `/` and
sqrt the GPU's -- none bit-comparable. The images agree as Monte-Carlo estimates of one integrand with one sample sequence: compare at
--validation-spp >= 64 (commands.sh does) with tools/reference_crosscheck/compare_images.py (RMSE, coverage) and bin/rptr_compare."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from realtimepathtracingresearchframework_amd import abi, scenes, vks  # noqa: E402


def config_scene(name, small):
    if name == "c1":
        return scenes.cornell32(), (256, 256), 1, 64
    if name == "c2g":
        s = scenes.grid(100, 50, name="grid-10k") if small else scenes.grid_1m()
        return s, ((480, 270) if small else (1920, 1080)), 4, 64
    if name == "c3":
        s = scenes.grid(100, 50, with_emitters=True, name="grid-10k-lights") if small else scenes.grid_1m_lights()
        return s, ((480, 270) if small else (1920, 1080)), 8, 64
    raise SystemExit("unknown configuration %r (c1, c2g, c3)" % name)


def sun_height_angle(d):
    """libapp/scene_state.h:79-80"""
    d = np.asarray(d, np.float64)
    d = d / np.linalg.norm(d)
    return 90.0 - np.degrees(np.arccos(d[1])), np.degrees(np.arctan2(d[2], d[0]))


def write_ini(path, scene_file, s, batch_spp, target_spp):
    """the reference's settings text (imstate.cpp:226-330); host/ini_config.hpp reads the same keys"""
    h, a = sun_height_angle(s.config.sun_dir)
    lc = abi.LightSamplingConfig.default()
    rp = abi.RenderParams.default()
    txt = ("[Application][]\n"
           "target spp= %d\nbatch spp= %d\nmax path depth= %d\nrr path depth= %d\nglossy-only mode= 0\npixel radius= %e\n"
           "[.][*output channel]\nOUTPUT_CHANNEL_COLOR= 1\n..\n[.][*variant]\n%s= 1\n..\n\n"
           "[Application][%s]\n[.][Sensor]\nlight bin size= %d\n..\n"
           "[.][Sun]\nheight= %e\nangle= %e\nturbidity= %e\nColor= %e %e %e\n..\n[.][Scene]\nbump scale= %e\n..\n"
           % (target_spp, batch_spp, rp.max_path_depth, rp.rr_path_depth, rp.pixel_radius, "PT_MEGAKERNEL", os.path.basename(scene_file),   # (vulkan/CMakeLists.txt:51 add_integrator(PT_MEGAKERNEL "megakernel" ...): the program's id)
              lc.bin_size, h, a, s.config.turbidity, s.config.albedo[0], s.config.albedo[1], s.config.albedo[2], s.config.bump_scale))
    open(path, "w").write(txt)


def export(name, out, small):
    s, (W, H), batch_spp, target_spp = config_scene(name, small)
    d = os.path.join(out, name)
    os.makedirs(d, exist_ok=True)
    scene_file = os.path.join(d, name + ".vks")
    names = vks.write_vks(scene_file, s, version=4)
    write_ini(os.path.join(d, name + ".ini"), scene_file, s, batch_spp, target_spp)
    back = vks.read_vks(scene_file)   # what a loader makes of the files: the materials as their textures decode
    cam = s.camera
    view = "--eye %g %g %g --center %g %g %g --up %g %g %g --fov %g" % (tuple(cam["eye"]) + tuple(cam["center"]) + tuple(cam["up"]) + (cam["fov"],))
    common = "%s.vks --config %s.ini --img %d %d %s --validation-spp %d --pfm" % (name, name, W, H, view, target_spp)
    open(os.path.join(d, "commands.sh"), "w").write(
        "#!/bin/bash\n# run inside this directory; RPTR = the reference's executable, RPTR_HIP = <repo>/realtimepathtracingresearchframework_amd/bin/rptr_hip,\n"
        "# SKY_DATA = <reference>/rendering/lights/sky_model_arhosek (the Hosek-Wilkie data headers: both programs fit the sky from the Sun settings)\n"
        "set -e\n"
        "${RPTR:?} --backend vulkan --disable-ui %s --validation ref\n"
        "${RPTR_HIP:?} %s --variant gltf --sky-data ${SKY_DATA:?} --validation hip\n"
        "python3 compare_images.py ref_%04d.pfm hip_%04d.pfm\n"
        "${RPTR_COMPARE:-$(dirname $RPTR_HIP)/rptr_compare} ref_%04d.pfm hip_%04d.pfm || true   # (1e-6 relative per value: the reference's regression notion, not expected to hold across GPUs)\n"
        % (common, common, target_spp, target_spp, target_spp, target_spp))
    import shutil
    shutil.copy(os.path.join(ROOT, "tools", "reference_crosscheck", "compare_images.py"), d)
    os.chmod(os.path.join(d, "commands.sh"), 0o755)
    json.dump({"configuration": name, "scene": s.name, "triangles": s.num_tris(), "instanced_triangles": s.num_instanced_tris(), "instances": len(s.instances),
               "materials": names, "textures": len(back.textures), "emitters": int(len(s.lights)), "image": [W, H], "batch_spp": batch_spp, "target_spp": target_spp,
               "camera": {k: [float(x) for x in np.atleast_1d(v)] for k, v in cam.items()},
               "sun": {"dir": list(s.config.sun_dir), "turbidity": s.config.turbidity, "albedo": list(s.config.albedo)}},
              open(os.path.join(d, "manifest.json"), "w"), indent=1)
    return d, s, back


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--out", default="crosscheck")
    ap.add_argument("--configs", default="c1,c2g,c3")
    ap.add_argument("--small", action="store_true")
    a = ap.parse_args()
    for name in a.configs.split(","):
        d, s, back = export(name.strip(), a.out, a.small)
        print("%s: %d triangles, %d materials (%d textures), %d emitters -> %s" % (name, s.num_tris(), len(s.materials), len(back.textures), len(s.lights), d))


if __name__ == "__main__":
    main()

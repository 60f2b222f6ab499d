"""tools/reference_crosscheck: the BASELINE configurations exported as files a real rptr loads (VERDICT r5 item 4b / missing 3: "an image
from the Vulkan reference" needs a Vulkan-RT GPU, GLM and GLFW, none of which this image has -- what can be prepared is that the run is one
script on a machine that has them). Checked here, on the CPU:
  * the exporter writes <name>.vks + textures + <name>.ini + commands.sh for C1 and C3 (at 1/100 of the triangle count);
  * the reference's own scene-file library (ext/libvkr/src/vkr.c compiled unmodified, oracle/_ref/libvkr_ref.so) reads those files to the
    values this repository's reader sees (skipped where the reference checkout was never there to build it);
  * what comes back is the procedural scene: quantised vertex / normal / uv streams bit for bit, the same instances, the same emitters --
    so camera rays hit the same triangles: the oracle's image of the read-back scene has the procedural scene's coverage (up to the
    format's 16-bit quaternions) and its radiance up to what the format cannot hold (5:6:5 colour blocks, 8-bit roughness, the loader's
    default normal texel);
  * the C++ host (bin/rptr_hip: host/vks_reader.hpp + host/ini_config.hpp) accepts the files and the .ini: spp, depth, variant, light bins,
    the Sun header; the command lines in commands.sh name flags both programs' parsers know (cmdline.cpp:296-474 / host/rptr_cli.cpp)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O
from realtimepathtracingresearchframework_amd import abi, build, scenes, vks
from test_vks import REF_LIB, _compare_with_dump, _ref, _same_streams

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "reference_crosscheck"))


@pytest.fixture(scope="module")
def exported(tmp_path_factory):
    import export_scenes
    out = str(tmp_path_factory.mktemp("crosscheck"))
    return {name: export_scenes.export(name, out, small=True) for name in ("c1", "c3")}


@pytest.mark.parametrize("name", ["c1", "c3"])
def test_exported_configuration_is_the_procedural_scene(exported, name):
    d, s, back = exported[name]
    path = os.path.join(d, name + ".vks")
    man = json.load(open(os.path.join(d, "manifest.json")))
    assert man["triangles"] == s.num_tris() and man["emitters"] == len(s.lights) and os.access(os.path.join(d, "commands.sh"), os.X_OK)
    if os.path.isfile(REF_LIB):   # the reference's reader on the exported files
        dump = os.path.join(d, "ref_dump.json")
        assert _ref().ref_vkr_dump(path.encode(), dump.encode()) == 0
        _compare_with_dump(path, json.load(open(dump)))
    _same_streams(s, back)
    assert len(back.instances) == len(s.instances) and len(back.lights) == len(s.lights)
    assert len(back.textures) == 3 * len(s.materials)   # the loader's three textures per material: nothing is left a literal
    back.camera, back.config, back.sky_key = s.camera, s.config, s.sky_key
    W, H, spp = (64, 64, 4) if name == "c1" else (96, 54, 8)
    a, sa = O.OracleScene(s).render(W, H, spp, variant=abi.VARIANT_GLTF)
    b, sb = O.OracleScene(back).render(W, H, spp, variant=abi.VARIANT_GLTF)
    # coverage: identical vertex streams and camera rays; the instance transforms come back through the format's 16-bit quaternion (even the
    # identity moves by 1e-4: vkr.c:1346-1411), which may move a silhouette across a sample in a pixel or two
    assert (a[..., 3] != b[..., 3]).mean() <= 2e-3 and np.abs(a[..., 3] - b[..., 3]).max() <= 0.25
    fa, fb = a[..., :3][np.isfinite(a[..., :3])], b[..., :3][np.isfinite(b[..., :3])]
    assert abs(float(fa.mean()) - float(fb.mean())) < 0.05 * float(fa.mean()) + 0.01


def test_cpp_host_accepts_the_exported_files_and_their_configuration(exported):
    exe = os.path.join(build.BIN_DIR, "rptr_hip")
    if not os.path.exists(exe):
        build.build_host_tools()
    d, s, _ = exported["c3"]
    out = subprocess.run([exe, "c3.vks", "--config", "c3.ini", "--describe"], cwd=d, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    scene_kv = dict(zip(lines[-2].split()[0::2], lines[-2].split()[1::2]))
    assert int(scene_kv["triangles"]) == s.num_tris() and int(scene_kv["lights"]) == len(s.lights) and int(scene_kv["materials"]) == len(s.materials)
    cfg = lines[-1].split()
    kv = dict(zip(cfg[1::2], cfg[2::2]))
    assert (int(kv["target_spp"]), int(kv["batch_spp"]), int(kv["max_path_depth"]), int(kv["bin_size"]), int(kv["variant"])) == (64, 8, 9, 16, abi.VARIANT_GLTF)
    assert int(kv["sun_changed"]) == 1
    cmds = open(os.path.join(d, "commands.sh")).read()
    for flag in ("--validation", "--validation-spp", "--pfm", "--img", "--eye", "--center", "--up", "--fov", "--config", "--backend vulkan", "--disable-ui"):
        assert flag in cmds

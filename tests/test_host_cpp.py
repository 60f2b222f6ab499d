"""The C-ABI header is valid C, the RenderBackend-shaped C++ host class compiles with g++ against it and links
to librptr_hip.so; without a GPU the demo fails loudly (exit 3, "no CPU fallback"), with one it renders."""
import os
import subprocess

import pytest

from realtimepathtracingresearchframework_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "realtimepathtracingresearchframework_amd", "host")


def _build_demo(tmp_path):
    if not os.path.exists(build.LIB_PATH):
        build.build_library()
    exe = str(tmp_path / "demo_host")
    libdir = os.path.dirname(build.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", os.path.join(HOST, "demo_host.cpp"), "-o", exe, "-L" + libdir, "-lrptr_hip",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_header_is_valid_c(tmp_path):
    src = tmp_path / "abi.c"
    src.write_text('#include "rptr_hip.h"\n#include "rptr_bvh.h"\n'
                   'int main(void){ return sizeof(RptrBaseMaterial)==80 && sizeof(RptrBvhNode)==64 && sizeof(RptrBvh4Node)==64 && sizeof(RptrBvhInstance)==128 && sizeof(RptrTriLightData)==48 ? 0 : 1; }\n')
    exe = str(tmp_path / "abi_c")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), str(src), "-o", exe])
    assert subprocess.call([exe]) == 0


def test_cpp_host_fails_loudly_without_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    exe = _build_demo(tmp_path)
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 3 and "no CPU fallback" in p.stderr


@pytest.mark.gpu
def test_cpp_host_renders_on_gpu(tmp_path):
    exe = _build_demo(tmp_path)
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert "mean radiance" in p.stdout
    # the adapter surface beyond frames: device-resident ray queries (enable_ray_queries / render_ray_queries), the RaytraceBackend-shaped
    # class, light_sampling_variant NONE
    assert "ray queries: device buffers = host arrays" in p.stdout and "same hits" in p.stdout and "light sampling NONE: image unchanged" in p.stdout, p.stdout
    # the reference's begin_frame / draw_frame / end_frame loop with a CommandStream (two frames in flight, statistics two frames late,
    # read-backs of the newest frame) gives the images of the synchronous loop (the adapter only reached the synchronous call)
    assert "frame loop through a CommandStream: images = the synchronous loop's; frame_stats_delay 2 / 0, first valid stats at frame 2 / 0, spp 8 / 8" in p.stdout, p.stdout

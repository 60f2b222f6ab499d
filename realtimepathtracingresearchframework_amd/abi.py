"""ctypes mirror of include/rptr_hip.h and include/rptr_bvh.h.

Plain data only; no compute. The struct layouts are bit-compatible with the
reference's shared C++/GLSL structs (file:line cited in include/rptr_hip.h).
"""
import ctypes as C

RPTR_OK = 0
RPTR_E_INVALID = -1
RPTR_E_NO_DEVICE = -2
RPTR_E_NOMEM = -3
RPTR_E_UNSUPPORTED = -4
RPTR_E_HIP = -5

VARIANT_GLTF = 0
ABI_VERSION = 5  # RPTR_HIP_ABI_VERSION (include/rptr_hip.h): rptr_hip_create refuses anything else
VARIANT_SIMPLE = 1
VARIANT_GLTF_TRANSMISSION = 2
MESH_DYNAMIC, MESH_SUBTLY_DYNAMIC = 1, 2  # RptrMeshDesc.dynamic = Mesh::flags (librender/mesh.h:44-47)
# RBO rng_variant (librender/render_params.glsl.h:34-37)
RNG_VARIANT_UNIFORM, RNG_VARIANT_BN, RNG_VARIANT_SOBOL, RNG_VARIANT_Z_SBL = 0, 1, 2, 3
RNG_VARIANT_NAMES = ("UNIFORM", "BN", "SOBOL", "Z_SBL")
SOBOL_TABLE_BYTES = (1024 * 32 + 256 * 256) * 4      # RPTR_SOBOL_TABLE_BYTES
BN_TABLE_MIN_BYTES = (256 * 256 + 128 * 128 * 8) * 4  # RPTR_BN_TABLE_MIN_BYTES
VARIANT_NAMES = ["wavefront-gltf", "wavefront-diffuse", "wavefront-gltf-transmission"]

BASE_MATERIAL_NOALPHA = 0x01
BASE_MATERIAL_ONESIDED = 0x02
BASE_MATERIAL_VOLUME = 0x04

MAX_PATH_DEPTH = 9
DEFAULT_RR_PATH_DEPTH = 2
BINNED_LIGHTS_BIN_MAX_SIZE = 16
RAY_EPSILON = 0.000005


class BaseMaterial(C.Structure):  # rendering/bsdfs/base_material.h.glsl:13-34
    _fields_ = [
        ("base_color", C.c_float * 3),
        ("normal_map", C.c_int32),
        ("flags", C.c_uint32),
        ("roughness", C.c_float),
        ("specular", C.c_float),
        ("metallic", C.c_float),
        ("sheen", C.c_float),
        ("sheen_tint", C.c_float),
        ("clearcoat", C.c_float),
        ("clearcoat_gloss", C.c_float),
        ("ior", C.c_float),
        ("specular_transmission", C.c_float),
        ("anisotropy", C.c_float),
        ("specular_tint", C.c_float),
        ("transmission_color", C.c_float * 3),
        ("emission_intensity", C.c_float),
    ]


def make_material(base_color=(0.9, 0.9, 0.9), roughness=1.0, specular=0.5, metallic=0.0, ior=1.5,
                  emission_intensity=0.0, flags=BASE_MATERIAL_NOALPHA):
    """BaseMaterial with the reference's defaults (base_material.h.glsl:14-33)."""
    m = BaseMaterial()
    m.base_color[:] = base_color
    m.normal_map = -1
    m.flags = flags
    m.roughness = roughness
    m.specular = specular
    m.metallic = metallic
    m.sheen = 0.0
    m.sheen_tint = 0.0
    m.clearcoat = 0.0
    m.clearcoat_gloss = 0.1
    m.ior = ior
    m.specular_transmission = 0.0
    m.anisotropy = 0.0
    m.specular_tint = 0.0
    m.transmission_color[:] = (1.0, 1.0, 1.0)
    m.emission_intensity = emission_intensity
    return m


class TriLightData(C.Structure):  # rendering/lights/tri.h.glsl:13-26
    _fields_ = [("v0", C.c_float * 3), ("v1", C.c_float * 3), ("v2", C.c_float * 3), ("radiance", C.c_float * 3)]


class RenderRayQuery(C.Structure):  # librender/render_params.glsl.h:165-170
    _fields_ = [("origin", C.c_float * 3), ("mode_or_data", C.c_int32), ("dir", C.c_float * 3), ("t_max", C.c_float)]


class LightSamplingConfig(C.Structure):  # librender/render_params.glsl.h:123-128
    _fields_ = [("light_mis_angle", C.c_float), ("bin_size", C.c_int32), ("min_perceived_receiver_dist", C.c_float),
                ("min_radiance", C.c_float)]

    @staticmethod
    def default():
        return LightSamplingConfig(0.0, 16, 15.0, 0.0)


class RenderParams(C.Structure):  # librender/render_params.glsl.h:130-155
    _fields_ = [
        ("batch_spp", C.c_int32), ("max_path_depth", C.c_int32), ("rr_path_depth", C.c_int32), ("glossy_only_mode", C.c_int32),
        ("aperture_radius", C.c_float), ("focus_distance", C.c_float), ("pixel_radius", C.c_float), ("variance_radius", C.c_float),
        ("output_channel", C.c_int32), ("output_moment", C.c_int32), ("exposure", C.c_float), ("early_tone_mapping_mode", C.c_int32),
        ("reprojection_mode", C.c_int32), ("spp_accumulation_window", C.c_int32), ("enable_raster_taa", C.c_int32),
        ("render_upscale_factor", C.c_int32),
        ("focal_length", C.c_float), ("_pad3", C.c_int32), ("_pad4", C.c_int32), ("_pad5", C.c_int32),
    ]

    @staticmethod
    def default():
        p = RenderParams()
        p.batch_spp = 1
        p.max_path_depth = MAX_PATH_DEPTH
        p.rr_path_depth = DEFAULT_RR_PATH_DEPTH
        p.glossy_only_mode = 0
        p.aperture_radius = 0.0
        p.focus_distance = 2.5
        p.pixel_radius = 1.0
        p.variance_radius = 4.0
        p.output_channel = 0
        p.output_moment = 0
        p.exposure = 0.0
        p.early_tone_mapping_mode = -1
        p.reprojection_mode = 0
        p.spp_accumulation_window = 8
        p.enable_raster_taa = 0
        p.render_upscale_factor = 1
        p.focal_length = 35.0
        return p


class SkyModelParams(C.Structure):  # rendering/lights/sky_model_arhosek/sky_model.h.glsl:7-10
    _fields_ = [("configs", (C.c_float * 4) * 9), ("radiances", C.c_float * 4)]


class SceneParams(C.Structure):
    _fields_ = [("sky_params", SkyModelParams), ("sun_dir", C.c_float * 3), ("sun_cos_angle", C.c_float),
                ("sun_radiance", C.c_float * 4), ("normal_z_scale", C.c_float), ("_pad", C.c_int32 * 3)]


class Camera(C.Structure):  # librender/render_backend.h:26-31
    _fields_ = [("pos", C.c_float * 3), ("dir", C.c_float * 3), ("up", C.c_float * 3), ("fovy", C.c_float)]


class GeometryDesc(C.Structure):
    _fields_ = [("qpos", C.c_void_p), ("qnrm_uv", C.c_void_p), ("num_tris", C.c_uint32), ("has_normals", C.c_uint32),
                ("has_uvs", C.c_uint32), ("quantized_scaling", C.c_float * 3), ("quantized_offset", C.c_float * 3)]


class MeshDesc(C.Structure):
    _fields_ = [("first_geometry", C.c_uint32), ("num_geometries", C.c_uint32), ("dynamic", C.c_uint32)]


class ParameterizedMeshDesc(C.Structure):
    _fields_ = [("mesh", C.c_uint32), ("material_offsets", C.c_void_p), ("tri_material_ids", C.c_void_p)]


class InstanceDesc(C.Structure):
    _fields_ = [("transform", C.c_float * 12), ("parameterized_mesh", C.c_uint32)]


class TextureDesc(C.Structure):  # include/rptr_hip.h RptrTextureDesc
    _fields_ = [("rgba8", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("srgb", C.c_uint32), ("mip_levels", C.c_uint32)]


def textured_param(texture_id, channel=0):
    """The float whose bits say "read this parameter from a texture" (rendering/bsdfs/texture_channel_mask.h): sign bit,
    channel in bits 29..30 (scalar parameters), texture index in bits 0..28. Returned as a Python float that converts back to
    exactly these bits when stored into a c_float (NaN payloads survive the float32 round trip only by luck, so this goes
    through struct pack/unpack of the float32 itself)."""
    import struct
    bits = 0x80000000 | ((channel & 3) << 29) | (texture_id & 0x1FFFFFFF)
    return struct.unpack("<f", struct.pack("<I", bits))[0]


def float_bits(value):
    """the bits of `value` as a float32 (to look at a material parameter that may be a texture handle)"""
    import struct
    return struct.unpack("<I", struct.pack("<f", value))[0]


def set_float_bits(carray, index, bits):
    """writes raw bits into element `index` of a ctypes float array (Python floats cannot carry every NaN payload)"""
    C.cast(carray, C.POINTER(C.c_uint32))[index] = bits & 0xFFFFFFFF


class SceneDesc(C.Structure):
    _fields_ = [
        ("geometries", C.POINTER(GeometryDesc)), ("num_geometries", C.c_uint32),
        ("meshes", C.POINTER(MeshDesc)), ("num_meshes", C.c_uint32),
        ("parameterized_meshes", C.POINTER(ParameterizedMeshDesc)), ("num_parameterized_meshes", C.c_uint32),
        ("instances", C.POINTER(InstanceDesc)), ("num_instances", C.c_uint32),
        ("materials", C.POINTER(BaseMaterial)), ("num_materials", C.c_uint32),
        ("lights", C.POINTER(TriLightData)), ("num_lights", C.c_uint32),
        ("textures", C.POINTER(TextureDesc)), ("num_textures", C.c_uint32),
    ]


class CreateInfo(C.Structure):
    _fields_ = [("device_ordinal", C.c_int32), ("rank", C.c_int32), ("world_size", C.c_int32), ("stripe_rows", C.c_int32),
                ("stream", C.c_void_p), ("frames_in_flight", C.c_int32), ("abi_version", C.c_int32), ("flags", C.c_uint32), ("_pad", C.c_uint32)]


CREATE_SET_HW_QUEUES = 1  # RPTR_CREATE_SET_HW_QUEUES


class Stats(C.Structure):
    _fields_ = [
        ("render_time_ms", C.c_float), ("extend_time_ms", C.c_float), ("connect_time_ms", C.c_float), ("shade_time_ms", C.c_float),
        ("rays_closest", C.c_uint64), ("rays_shadow", C.c_uint64), ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64),
        ("hits_shaded", C.c_uint64), ("nodes_closest", C.c_uint64), ("tris_closest", C.c_uint64),
        ("spp", C.c_int32), ("launches_extend", C.c_int32), ("launches_connect", C.c_int32), ("_pad", C.c_int32),
        ("device_bytes_allocated", C.c_uint64),
        ("shade_only_time_ms", C.c_float), ("tail_time_ms", C.c_float), ("resolve_time_ms", C.c_float), ("_pad2", C.c_float),
    ]


class BvhNode(C.Structure):  # include/rptr_bvh.h
    _fields_ = [("lo0", C.c_float * 3), ("hi0", C.c_float * 3), ("lo1", C.c_float * 3), ("hi1", C.c_float * 3),
                ("child0", C.c_int32), ("child1", C.c_int32), ("cnt0", C.c_int32), ("cnt1", C.c_int32)]


class BvhTri(C.Structure):
    _fields_ = [("v0", C.c_float * 3), ("e1", C.c_float * 3), ("e2", C.c_float * 3), ("prim", C.c_uint32), ("geom", C.c_uint32),
                ("_pad", C.c_uint32)]


class BvhInstance(C.Structure):
    _fields_ = [("world_to_object", C.c_float * 12), ("object_to_world", C.c_float * 12), ("blas_root", C.c_int32),
                ("geometry_base", C.c_int32), ("instance_id", C.c_int32), ("flags", C.c_int32), ("_pad", C.c_int32 * 4)]


assert C.sizeof(BaseMaterial) == 80
assert C.sizeof(TriLightData) == 48
assert C.sizeof(RenderRayQuery) == 32
assert C.sizeof(RenderParams) == 80
assert C.sizeof(LightSamplingConfig) == 16
assert C.sizeof(SkyModelParams) == 160
assert C.sizeof(BvhNode) == 64
assert C.sizeof(BvhTri) == 48
assert C.sizeof(BvhInstance) == 128

# every symbol include/rptr_hip.h declares (checked by tests/test_abi.py)
COMM_ID_BYTES = 128  # RPTR_COMM_ID_BYTES
COMM_IPC_BYTES = 256  # RPTR_COMM_IPC_BYTES

EXPORTED_SYMBOLS = [
    "rptr_hip_create", "rptr_hip_abi_version", "rptr_hip_build_id", "rptr_hip_bvh_build_info", "rptr_hip_traversal_preset", "rptr_hip_destroy", "rptr_hip_last_error", "rptr_hip_name", "rptr_hip_set_stream",
    "rptr_hip_initialize", "rptr_hip_set_scene", "rptr_hip_update_vertices", "rptr_hip_update_vertices_device", "rptr_hip_refit", "rptr_hip_set_params",
    "rptr_hip_render", "rptr_hip_render_async", "rptr_hip_render_batch_async", "rptr_hip_render_batch_cameras_async", "rptr_hip_wait", "rptr_hip_set_stage_timing", "rptr_hip_set_freeze_frame", "rptr_hip_set_option", "rptr_hip_get_option", "rptr_hip_option_count", "rptr_hip_option_name", "rptr_hip_set_rng_variant", "rptr_hip_set_bvh_policy", "rptr_hip_bvh_rebuild_count", "rptr_hip_get_framebuffer_size", "rptr_hip_readback_f32", "rptr_hip_readback_u8", "rptr_hip_readback_aov",
    "rptr_hip_tile_rows", "rptr_hip_local_pixel_count", "rptr_hip_copy_tile_to_device", "rptr_hip_trace", "rptr_hip_trace_device", "rptr_hip_enable_ray_queries", "rptr_hip_render_ray_queries", "rptr_hip_set_light_sampling_variant", "rptr_hip_trace_counted",
    "rptr_hip_export_bvh", "rptr_hip_build_bvh_host", "rptr_hip_stats",
    "rptr_hip_comm_get_unique_id", "rptr_hip_comm_init_rank", "rptr_hip_comm_init_all", "rptr_hip_comm_destroy", "rptr_hip_comm_transport", "rptr_hip_comm_ipc_export", "rptr_hip_comm_ipc_init", "rptr_hip_gather", "rptr_hip_gather_all", "rptr_hip_gather_batch", "rptr_hip_gather_all_batch", "rptr_hip_readback_gathered_frame_f32",
    "rptr_hip_gathered_frame", "rptr_hip_readback_gathered_f32", "rptr_hip_comm_stats",
]

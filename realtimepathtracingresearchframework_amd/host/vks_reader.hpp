// vks_reader.hpp -- `.vks` scenes, `.vkt` textures and material parameter files of the reference's asset format, read in C++ into
// the flat scene the C ABI consumes (SceneDump -> RptrSceneDesc): what `Scene::load_vkrs` does with libvkr
// (librender/scene.cpp:544-977; ext/libvkr/src/vkr.c:771-1145 scene, :216-306 texture, :412-452 parameter files, :1382-1411
// transforms). The C++ twin of vks.py (reader half); tests compare the two on the same files, and vks.py itself is pinned against
// libvkr compiled from the reference checkout (tests/test_vks.py).
//
// As in vks.py: versions 1-2 (legacy single-mesh files) are not read; 16-bit per-triangle material ids keep their low byte (what the
// reference's backend uploads); index buffers are skipped (the vertex streams of a .vks file are unrolled); a level of detail other than the
// base level is chosen at load time (remove_first_lods); animated transforms: one frame.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "lights.hpp"
#include "scene_dump.hpp"
#include "sky_params.hpp"

namespace rptr {
namespace vks {

enum { VKR_MAGIC = 0xABCABC, VKT_MAGIC = 0xBC1BC1, QUANTIZED_TRANSFORM_SIZE = 24, MESH_FLAGS_INDICES = 0x1 }; // vkr.c:40,46; vkr.h:15,184
// VkFormat values a .vkt may carry (vkr.h:52-69)
enum { FMT_RGBA8_UNORM = 37, FMT_RGBA8_SRGB = 43, FMT_BC1_RGB_UNORM = 131, FMT_BC1_RGB_SRGB = 132, FMT_BC1_RGBA_UNORM = 133, FMT_BC1_RGBA_SRGB = 134,
       FMT_BC3_UNORM = 137, FMT_BC3_SRGB = 138, FMT_BC5_UNORM = 141 };

struct Error : std::runtime_error { // the reference's throw_error on a VkrResult != VKR_SUCCESS (scene.cpp:548-557)
    using std::runtime_error::runtime_error;
};

inline std::vector<uint8_t> read_file(const std::string &path, bool *exists = nullptr) {
    std::FILE *f = std::fopen(path.c_str(), "rb");
    if (exists) *exists = f != nullptr;
    if (!f) {
        if (exists) return {};
        throw Error("cannot open " + path);
    }
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> raw((size_t)n);
    const size_t got = n ? std::fread(raw.data(), 1, (size_t)n, f) : 0;
    std::fclose(f);
    if (got != (size_t)n) throw Error("short read: " + path);
    return raw;
}

// ------------------------------------------------------------------ transforms
// vkr_dequantize_transform (vkr.c:1382-1411): 24 bytes = translation (3 floats), signed uniform scale, quaternion (4 x u16) -> float[4][3]
inline void dequantize_transform(const uint8_t *raw, float m[4][3]) {
    float t[3], scaling;
    uint16_t qq[4];
    std::memcpy(t, raw, 12);
    std::memcpy(&scaling, raw + 12, 4);
    std::memcpy(qq, raw + 16, 8);
    float q[4];
    for (int i = 0; i < 4; ++i) q[i] = float(qq[i]) * (2.0f / float(0xFFFF)) - 1.0f;
    q[3] = -q[3];
    const float xx = q[0] * q[0], xy = q[0] * q[1], xz = q[0] * q[2], xw = q[0] * q[3];
    const float yy = q[1] * q[1], yz = q[1] * q[2], yw = q[1] * q[3];
    const float zz = q[2] * q[2], zw = q[2] * q[3];
    const float r[3][3] = {{1.0f - 2.0f * (yy + zz), 2.0f * (xy - zw), 2.0f * (xz + yw)},
                           {2.0f * (xy + zw), 1.0f - 2.0f * (xx + zz), 2.0f * (yz - xw)},
                           {2.0f * (xz - yw), 2.0f * (yz + xw), 1.0f - 2.0f * (xx + yy)}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) m[i][j] = r[i][j] * scaling;
    for (int j = 0; j < 3; ++j) m[3][j] = t[j];
}
// vkr_quantize_transform (vkr.c:1267-1379): float[4][3] (three basis vectors, then the translation) -> the 24 bytes; version 3 files
// store the floats and the reader quantises them into the table (vkr.c:1027-1035)
inline void quantize_transform(const float m[4][3], uint8_t out[24]) {
    float scaling = 0.0f;
    for (int i = 0; i < 3; ++i) scaling += m[0][i] * m[0][i];
    scaling = std::sqrt(scaling);
    const float det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
                      m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
    if (det < 0.0f) scaling = -scaling;
    float a[3][3];
    const float inv_s = 1.0f / scaling;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) a[i][j] = m[i][j] * inv_s;
    float q[4];
    if (a[0][0] + a[1][1] + a[2][2] > 0.1f) {
        q[0] = a[2][1] - a[1][2];
        q[1] = a[0][2] - a[2][0];
        q[2] = a[1][0] - a[0][1];
        q[3] = 1.0f + a[0][0] + a[1][1] + a[2][2];
    } else if (a[0][0] > a[1][1] && a[0][0] > a[2][2]) {
        q[0] = 1.0f + a[0][0] - a[1][1] - a[2][2];
        q[1] = a[1][0] + a[0][1];
        q[2] = a[0][2] + a[2][0];
        q[3] = a[2][1] - a[1][2];
    } else if (a[1][1] > a[0][0] && a[1][1] > a[2][2]) {
        q[0] = a[1][0] + a[0][1];
        q[1] = 1.0f + a[1][1] - a[0][0] - a[2][2];
        q[2] = a[2][1] + a[1][2];
        q[3] = a[0][2] - a[2][0];
    } else {
        q[0] = a[0][2] + a[2][0];
        q[1] = a[2][1] + a[1][2];
        q[2] = 1.0f + a[2][2] - a[0][0] - a[1][1];
        q[3] = a[1][0] - a[0][1];
    }
    float length_sq = 0.0f;
    for (int i = 0; i < 4; ++i) length_sq += q[i] * q[i];
    const float inv = 1.0f / std::sqrt(length_sq);
    for (int i = 0; i < 4; ++i) q[i] *= inv;
    q[3] = -q[3];
    uint16_t qq[4];
    for (int i = 0; i < 4; ++i) qq[i] = (uint16_t)((long long)std::floor((q[i] * 0.5f + 0.5f) * float(0xFFFF) - 0.5f) & 0xFFFF);
    std::memcpy(out, m[3], 12);
    std::memcpy(out + 12, &scaling, 4);
    std::memcpy(out + 16, qq, 8);
}
// AnimationData::dequantize (librender/scene.cpp:22-41): vks_flip * mat4(tx) as the row-major 3x4 object-to-world matrix; the columns
// of tx are the rows of the float[4][3]; vks_flip = rows (-x, z, y)
inline void instance_transform(const uint8_t *raw, float out[12]) {
    float m[4][3];
    dequantize_transform(raw, m);
    static const float flip[3][3] = {{-1.f, 0.f, 0.f}, {0.f, 0.f, 1.f}, {0.f, 1.f, 0.f}};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) // a matrix product as the reference computes it, not a swizzle: -1 * 0 + 0 + 0 is +0, not -0
            out[r * 4 + c] = (flip[r][0] * m[c][0] + flip[r][1] * m[c][1]) + flip[r][2] * m[c][2];
}

// ------------------------------------------------------------------ block compression (published BC1 / BC3 / BC5 block layouts)
inline void expand565(uint32_t c, double rgb[3]) {
    const uint32_t r = (c >> 11) & 31u, g = (c >> 5) & 63u, b = c & 31u;
    rgb[0] = double((r << 3) | (r >> 2));
    rgb[1] = double((g << 2) | (g >> 4));
    rgb[2] = double((b << 3) | (b >> 2));
}
// 8-byte colour block -> 16 RGBA texels in [0, 255]; three-colour mode (c0 <= c1, unless force4): entry 3 is transparent black
inline void decode_bc1_block(const uint8_t *b, bool force4, double out[16][4]) {
    const uint32_t c0 = b[0] | (uint32_t(b[1]) << 8), c1 = b[2] | (uint32_t(b[3]) << 8);
    double e0[3], e1[3], pal[4][4];
    expand565(c0, e0);
    expand565(c1, e1);
    const bool four = force4 || c0 > c1;
    for (int k = 0; k < 3; ++k) {
        pal[0][k] = e0[k];
        pal[1][k] = e1[k];
        pal[2][k] = four ? (2 * e0[k] + e1[k]) / 3.0 : (e0[k] + e1[k]) / 2.0;
        pal[3][k] = four ? (e0[k] + 2 * e1[k]) / 3.0 : 0.0;
    }
    pal[0][3] = pal[1][3] = pal[2][3] = 255.0;
    pal[3][3] = four ? 255.0 : 0.0;
    const uint32_t bits = b[4] | (uint32_t(b[5]) << 8) | (uint32_t(b[6]) << 16) | (uint32_t(b[7]) << 24);
    for (int t = 0; t < 16; ++t)
        for (int k = 0; k < 4; ++k) out[t][k] = pal[(bits >> (2 * t)) & 3u][k];
}
inline void decode_bc4_block(const uint8_t *b, double out[16]) {
    const double a0 = b[0], a1 = b[1];
    uint64_t bits = 0;
    for (int k = 0; k < 6; ++k) bits |= uint64_t(b[2 + k]) << (8 * k);
    double pal[8];
    pal[0] = a0;
    pal[1] = a1;
    const bool eight = a0 > a1;
    for (int i = 1; i < 7; ++i) pal[1 + i] = eight ? ((7 - i) * a0 + i * a1) / 7.0 : (i < 5 ? ((5 - i) * a0 + i * a1) / 5.0 : 0.0);
    if (!eight) {
        pal[6] = 0.0;
        pal[7] = 255.0;
    }
    for (int t = 0; t < 16; ++t) out[t] = pal[(bits >> (3 * t)) & 7u];
}
// level 0 of a .vkt payload -> width * height RGBA8
inline std::vector<uint8_t> decode_texture(const uint8_t *data, size_t size, int width, int height, int fmt) {
    std::vector<uint8_t> img((size_t)width * height * 4);
    if (fmt == FMT_RGBA8_UNORM || fmt == FMT_RGBA8_SRGB) {
        if (size < img.size()) throw Error("texture payload too short");
        std::memcpy(img.data(), data, img.size());
        return img;
    }
    const int bw = (width + 3) / 4, bh = (height + 3) / 4;
    const bool bc1 = fmt == FMT_BC1_RGB_UNORM || fmt == FMT_BC1_RGB_SRGB || fmt == FMT_BC1_RGBA_UNORM || fmt == FMT_BC1_RGBA_SRGB;
    const bool bc3 = fmt == FMT_BC3_UNORM || fmt == FMT_BC3_SRGB, bc5 = fmt == FMT_BC5_UNORM;
    if (!bc1 && !bc3 && !bc5) throw Error("unsupported texture format " + std::to_string(fmt));
    const size_t block_bytes = bc1 ? 8 : 16;
    if (size < (size_t)bw * bh * block_bytes) throw Error("texture payload too short");
    for (int by = 0; by < bh; ++by)
        for (int bx = 0; bx < bw; ++bx) {
            const uint8_t *b = data + ((size_t)by * bw + bx) * block_bytes;
            double tex[16][4];
            if (bc5) {
                double r[16], g[16];
                decode_bc4_block(b, r);
                decode_bc4_block(b + 8, g);
                for (int t = 0; t < 16; ++t) {
                    tex[t][0] = r[t];
                    tex[t][1] = g[t];
                    tex[t][2] = 0.0;
                    tex[t][3] = 255.0;
                }
            } else {
                decode_bc1_block(bc3 ? b + 8 : b, bc3, tex);
                if (bc3) {
                    double a[16];
                    decode_bc4_block(b, a);
                    for (int t = 0; t < 16; ++t) tex[t][3] = a[t];
                } else if (fmt == FMT_BC1_RGB_UNORM || fmt == FMT_BC1_RGB_SRGB)
                    for (int t = 0; t < 16; ++t) tex[t][3] = 255.0; // VK_FORMAT_BC1_RGB_*: the transparent entry decodes as opaque black
            }
            for (int t = 0; t < 16; ++t) {
                const int x = bx * 4 + (t & 3), y = by * 4 + (t >> 2);
                if (x >= width || y >= height) continue;
                for (int k = 0; k < 4; ++k) {
                    const double v = std::floor(tex[t][k] + 0.5);
                    img[((size_t)y * width + x) * 4 + k] = (uint8_t)(v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v));
                }
            }
        }
    return img;
}

struct Texture {
    bool present = false;
    int width = 0, height = 0, format = 0;
    uint32_t levels = 0;       // mip levels in `rgba` (RptrTextureDesc.mip_levels: 0 = level 0 only)
    std::vector<uint8_t> rgba; // the levels back to back as RGBA8
};
// -> every level of the file as RGBA8 (the reference maps dataSize bytes from dataOffset and uploads every level: scene.cpp:866,
// vulkan/resource_utils.cpp:86-100); present = false when the file does not exist (textures are optional, vkr.c:475-489)
inline Texture read_vkt(const std::string &path) {
    Texture t;
    bool exists = false;
    const std::vector<uint8_t> raw = read_file(path, &exists);
    if (!exists) return t;
    if (raw.size() < 32) throw Error(path + " is not a .vkt file.");
    int32_t h6[6];
    uint64_t size;
    std::memcpy(h6, raw.data(), 24);
    std::memcpy(&size, raw.data() + 24, 8);
    if (h6[0] != VKT_MAGIC) throw Error(path + " is not a .vkt file.");
    if (h6[1] != 1) throw Error("Unsupported file version " + std::to_string(h6[1]) + " in " + path);
    const int nmips = h6[2];
    if (nmips < 1 || raw.size() < 32 + (size_t)24 * nmips) throw Error("Failed to read mip level header.");
    size_t at = 32 + (size_t)24 * nmips; // dataOffset = ftell(f) after the mip headers
    t.format = h6[5];
    int pw = 0, ph = 0;
    for (int l = 0; l < nmips; ++l) {
        int32_t mw, mh;
        uint64_t msize;
        std::memcpy(&mw, raw.data() + 32 + (size_t)24 * l, 4);
        std::memcpy(&mh, raw.data() + 36 + (size_t)24 * l, 4);
        std::memcpy(&msize, raw.data() + 40 + (size_t)24 * l, 8);
        if (l && (mw != std::max(1, pw / 2) || mh != std::max(1, ph / 2)))
            throw Error("mip level " + std::to_string(l) + " of " + path + " is " + std::to_string(mw) + " x " + std::to_string(mh));
        if (at + msize > raw.size()) throw Error("texture payload too short");
        const std::vector<uint8_t> level = decode_texture(raw.data() + at, (size_t)msize, mw, mh, t.format);
        t.rgba.insert(t.rgba.end(), level.begin(), level.end());
        at += (size_t)msize;
        if (l == 0) t.width = mw, t.height = mh;
        pw = mw, ph = mh;
    }
    t.present = true;
    t.levels = nmips > 1 ? (uint32_t)nmips : 0u;
    return t;
}

// ------------------------------------------------------------------ material parameter files (vkr.c:412-452): one float per line
inline bool read_params(const std::string &path, size_t max_values, std::vector<float> &vals) {
    bool exists = false;
    const std::vector<uint8_t> raw = read_file(path, &exists);
    vals.clear();
    if (!exists) return false;
    std::string text(raw.begin(), raw.end());
    size_t at = 0;
    while (at <= text.size() && vals.size() < max_values) {
        size_t nl = text.find('\n', at);
        if (nl == std::string::npos) nl = text.size();
        std::string line = text.substr(at, nl - at);
        while (!line.empty() && (line.back() == '\r' || line.back() == ' ' || line.back() == '\t')) line.pop_back();
        size_t b = 0;
        while (b < line.size() && (line[b] == ' ' || line[b] == '\t')) ++b;
        line = line.substr(b);
        if (line.empty()) break;
        char *end = nullptr;
        const float v = std::strtof(line.c_str(), &end);
        if (end == line.c_str() || *end != 0) throw Error("Invalid number format in vkr_parse_material_param_file");
        vals.push_back(v);
        at = nl + 1;
    }
    return true;
}
inline std::string texture_dir(const std::string &scene_file) { // buildTextureDir, vkr.c:80-110
    const size_t dot = scene_file.rfind('.');
    return (dot == std::string::npos ? scene_file : scene_file.substr(0, dot)) + "_textures/";
}

// ------------------------------------------------------------------ header (vkr_open_scene, vkr.c:771-1145), versions 3 and 4
struct MeshHeader {
    float vertexScale[3], vertexOffset[3];
    uint32_t flags = 0;
    uint64_t vertexBufferOffset = 0, numSegments = 0, numTriangles = 0;
    int32_t materialIdBufferBase = 0;
    uint32_t numMaterialsInRange = 0;
    int64_t lodGroup = 0;
    std::vector<uint64_t> segmentNumTriangles;
    std::vector<int32_t> segmentMaterialBaseOffsets;
    std::string name;
    uint64_t normalUvBufferOffset = 0, materialIdBufferOffset = 0, indexBufferOffset = 0;
    int materialIdSize = 1;
};
struct InstanceHeader {
    int32_t meshId = 0;
    uint32_t flags = 0, transformIndex = 0;
};
struct LodGroup {
    std::vector<int64_t> meshIds;
    std::vector<float> detailReduction;
};
struct Header {
    int version = 0;
    uint64_t numMeshes = 0, numInstances = 0, numMaterials = 0, numTriangles = 0, numLodGroups = 1;
    uint64_t numFrames = 1, numStaticTransforms = 0, numAnimatedTransforms = 0;
    std::vector<MeshHeader> meshes;
    std::vector<InstanceHeader> instances;
    std::vector<LodGroup> lodGroups;
    std::vector<std::string> materialNames;
    std::vector<uint8_t> transforms; // the quantised transform table
    std::vector<uint8_t> raw;
};

class Cursor {
public:
    Cursor(const std::vector<uint8_t> &raw, const std::string &name) : raw_(raw), name_(name) {}
    size_t at = 0;
    template <class T>
    T take() {
        if (at + sizeof(T) > raw_.size()) throw Error("Failed to read header structure from " + name_ + ".");
        T v;
        std::memcpy(&v, raw_.data() + at, sizeof(T));
        at += sizeof(T);
        return v;
    }
    std::string string() { // vkr_load_string: u64 length, then length + 1 bytes
        const uint64_t n = take<uint64_t>();
        if (at + n + 1 > raw_.size()) throw Error("Failed to read string from " + name_ + ".");
        std::string s((const char *)raw_.data() + at, (size_t)n);
        at += (size_t)n + 1;
        return s;
    }

private:
    const std::vector<uint8_t> &raw_;
    std::string name_;
};

inline Header read_header(const std::string &path) {
    Header v;
    v.raw = read_file(path);
    Cursor c(v.raw, path);
    if (v.raw.size() < 8 || c.take<int32_t>() != VKR_MAGIC) throw Error(path + " is not a .vks file.");
    v.version = c.take<int32_t>();
    if (v.version < 3 || v.version > 4) throw Error("Unsupported version " + std::to_string(v.version) + " in " + path + ".");
    (void)c.take<uint64_t>(); // flags
    const uint64_t headerSize = c.take<uint64_t>(), dataOffset = c.take<uint64_t>();
    if (!(headerSize > 0 && dataOffset >= headerSize)) throw Error("Failed to read header size & data offset from " + path + ".");
    v.numMeshes = c.take<uint64_t>();
    v.numInstances = c.take<uint64_t>();
    v.numMaterials = c.take<uint64_t>();
    v.numTriangles = c.take<uint64_t>();
    const uint64_t groups = c.take<uint64_t>();
    int64_t lod_offset = 0, animationOffset = 0;
    if (v.version >= 4) {
        v.numLodGroups = c.take<uint64_t>();
        lod_offset = c.take<int64_t>();
        (void)c.take<uint64_t>(); // numBoneIndexTuples
        (void)c.take<int64_t>();  // boneIndexTuplesOffset
        (void)c.take<float>();    // animationStart
        (void)c.take<float>();    // animationStep
        v.numFrames = c.take<uint64_t>();
        v.numStaticTransforms = c.take<uint64_t>();
        v.numAnimatedTransforms = c.take<uint64_t>();
        animationOffset = c.take<int64_t>();
    } else {
        v.numFrames = 1;
        v.numStaticTransforms = v.numInstances;
    }
    if (v.numMeshes == 0 || v.numInstances == 0 || groups == 0 || v.numLodGroups == 0) throw Error("Failed to read valid object counts from " + path + ".");
    if (headerSize != c.at) throw Error("Mismatching header size in " + path + ".");
    for (uint64_t i = 0; i < v.numMeshes; ++i) {
        MeshHeader m;
        for (float &x : m.vertexScale) x = c.take<float>();
        for (float &x : m.vertexOffset) x = c.take<float>();
        m.flags = (uint32_t)c.take<uint64_t>();
        const uint64_t header_end = c.take<uint64_t>();
        m.vertexBufferOffset = c.take<uint64_t>();
        m.numSegments = c.take<uint64_t>();
        m.numTriangles = c.take<uint64_t>();
        m.materialIdBufferBase = c.take<int32_t>();
        m.numMaterialsInRange = c.take<uint32_t>();
        int reserved = 5;
        if (v.version >= 4) {
            m.lodGroup = c.take<int64_t>();
            --reserved;
        }
        for (int r = 0; r < reserved; ++r) (void)c.take<uint64_t>();
        if (m.lodGroup >= (int64_t)v.numLodGroups) throw Error("Invalid LoD group specified for mesh " + std::to_string(i) + " from " + path + ".");
        for (uint64_t s = 0; s < m.numSegments; ++s) m.segmentNumTriangles.push_back(c.take<uint64_t>());
        for (uint64_t s = 0; s < m.numSegments; ++s) m.segmentMaterialBaseOffsets.push_back(c.take<int32_t>());
        m.name = c.string();
        if (header_end != c.at) throw Error("Mismatching header offset for mesh " + std::to_string(i) + " from " + path + ".");
        v.meshes.push_back(m);
    }
    std::vector<uint8_t> legacy;
    for (uint64_t g = 0; g < groups; ++g) {
        const uint32_t iflags = c.take<uint32_t>();
        const int32_t mesh_id = c.take<int32_t>();
        const uint64_t header_end = c.take<uint64_t>(), data_offset = c.take<uint64_t>(), count = c.take<uint64_t>();
        (void)c.string();
        if (data_offset != c.at) throw Error("Mismatching data offset for instance group " + std::to_string(g) + " from " + path + ".");
        for (uint64_t k = 0; k < count; ++k) {
            InstanceHeader in;
            in.meshId = mesh_id;
            in.flags = iflags;
            if (v.version >= 4)
                in.transformIndex = c.take<uint32_t>();
            else { // version 3 stores the float[4][3] itself
                float m[4][3];
                for (int a = 0; a < 4; ++a)
                    for (int b = 0; b < 3; ++b) m[a][b] = c.take<float>();
                uint8_t q[24];
                quantize_transform(m, q);
                in.transformIndex = (uint32_t)(legacy.size() / QUANTIZED_TRANSFORM_SIZE);
                legacy.insert(legacy.end(), q, q + 24);
            }
            v.instances.push_back(in);
        }
        if (header_end != c.at) throw Error("Mismatching header offset for instance group " + std::to_string(g) + " from " + path + ".");
    }
    if (v.instances.size() != v.numInstances) throw Error("Failed to read valid object counts from " + path + ".");
    if (v.version >= 4) {
        if ((uint64_t)lod_offset != c.at) throw Error("Read invalid LoD group offset from " + path + ".");
        for (uint64_t g = 0; g < v.numLodGroups; ++g) {
            LodGroup lg;
            const uint64_t n = c.take<uint64_t>();
            for (uint64_t k = 0; k < n; ++k) lg.meshIds.push_back(c.take<int64_t>());
            for (uint64_t k = 0; k < n; ++k) lg.detailReduction.push_back(c.take<float>());
            v.lodGroups.push_back(lg);
        }
    } else
        v.lodGroups.push_back(LodGroup{});
    if (dataOffset != c.at) throw Error("Mismatching body data offset " + path + ".");
    for (uint64_t i = 0; i < v.numMaterials; ++i) v.materialNames.push_back(c.string());
    uint64_t offset = c.at;
    for (size_t i = 0; i < v.meshes.size(); ++i) { // vkr.c:1110-1138
        MeshHeader &m = v.meshes[i];
        if (m.vertexBufferOffset != offset) throw Error("Mismatching data offset for mesh " + std::to_string(i) + " from " + path + ".");
        const uint64_t n = m.numTriangles;
        offset += 24 * n;
        m.normalUvBufferOffset = offset;
        offset += 24 * n;
        m.materialIdBufferOffset = offset;
        m.materialIdSize = (m.numMaterialsInRange <= 0x100 || m.numSegments > 1) ? 1 : 2;
        offset += (uint64_t)m.materialIdSize * n;
        if (m.flags & MESH_FLAGS_INDICES) {
            m.indexBufferOffset = offset;
            offset += 12 * n;
        }
    }
    if (v.version < 4) {
        v.transforms = legacy;
        return v;
    }
    const uint64_t n_tf = v.numStaticTransforms + v.numFrames * v.numAnimatedTransforms;
    if (animationOffset <= 0 || (uint64_t)animationOffset + n_tf * QUANTIZED_TRANSFORM_SIZE > v.raw.size())
        throw Error("Failed to read the transform table from " + path + ".");
    v.transforms.assign(v.raw.begin() + animationOffset, v.raw.begin() + animationOffset + (long)(n_tf * QUANTIZED_TRANSFORM_SIZE));
    return v;
}

inline uint64_t transform_offset(uint64_t index, uint64_t num_static, uint64_t num_animated, uint64_t frame) { // vkr_get_transform_offset, vkr.c:197-208
    return index < num_static ? index : num_static + (index - num_static) + frame * num_animated;
}

// ------------------------------------------------------------------ materials (vkr_load_material, vkr.c:509-620; scene.cpp:818-975)
inline RptrBaseMaterial default_material() { // base_material.h.glsl:14-33
    RptrBaseMaterial m;
    std::memset(&m, 0, sizeof(m));
    m.base_color[0] = m.base_color[1] = m.base_color[2] = 0.9f;
    m.normal_map = -1;
    m.roughness = 1.0f;
    m.specular = 0.5f;
    m.clearcoat_gloss = 0.1f;
    m.ior = 1.5f;
    m.transmission_color[0] = m.transmission_color[1] = m.transmission_color[2] = 1.0f;
    return m;
}
inline float textured_param(uint32_t texture_id, uint32_t channel) { // rendering/bsdfs/texture_channel_mask.h: sign bit, channel 29..30, texture index
    const uint32_t bits = 0x80000000u | ((channel & 3u) << 29) | (texture_id & 0x1FFFFFFFu);
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

// Scene::load_vkrs for one file (without its override parameters): one mesh + parameterized mesh per .vks mesh (a geometry per segment,
// per-triangle material ids when a single segment spans several materials), base-LoD instances, per material the three standard
// textures (1 x 1 defaults when a file is missing) wired in as texture handles; then emitters, default camera, default sky.
// SceneLoaderParams::PerFile (librender/scene.h:33-45), the members that change what is rendered; `dynamic_meshes` = a reference build
// with ENABLE_DYNAMIC_MESHES (scene.cpp:691-706)
struct LoadParams {
    bool ignore_textures = false, load_specularity = false;
    uint64_t frame = 0;
    int remove_first_lods = 0;
    float instance_pruning_probability = 0.0f;
    bool small_deformation = false, ignore_animation = false, dynamic_meshes = true;
};

inline float halton2(uint32_t index) { // util/compute_util.h:19-33: reversed bits as a float's mantissa
    uint32_t r = 0;
    for (int b = 0; b < 32; ++b) r |= ((index >> b) & 1u) << (31 - b);
    const uint32_t bits = 0x3f800000u | (r >> 9);
    float f;
    std::memcpy(&f, &bits, 4);
    return f - 1.0f;
}

inline std::string extended_material_name(const std::string &tex_dir, const std::string &name) { // vkr.c:526-534
    bool exists = false;
    const std::vector<uint8_t> ex = read_file(tex_dir + name + "_Ex.txt", &exists);
    return exists ? std::string(ex.begin(), ex.end()) : name;
}

inline SceneDump read_scene(const std::string &path, const std::string &data_dir, const LoadParams &lp) {
    const bool ignore_textures = lp.ignore_textures, load_specularity = lp.load_specularity;
    const uint64_t frame = lp.frame;
    const int remove_first_lods = lp.remove_first_lods;
    const Header v = read_header(path);
    SceneDump s;
    const std::string tex_dir = texture_dir(path);
    std::vector<std::string> extended_names;
    for (const std::string &name : v.materialNames) extended_names.push_back(extended_material_name(tex_dir, name));
    size_t n_geom = 0;
    for (const MeshHeader &m : v.meshes)
        for (uint64_t n : m.segmentNumTriangles) n_geom += n ? 1 : 0;
    s.qpos.reserve(n_geom); // (pointers into these vectors are handed out below)
    s.qnu.reserve(n_geom);
    s.offsets.resize(v.meshes.size());
    s.tri_ids.resize(v.meshes.size());
    for (size_t i = 0; i < v.meshes.size(); ++i) {
        const MeshHeader &vm = v.meshes[i];
        const uint32_t first = (uint32_t)s.geometries.size();
        uint64_t base = 0;
        for (uint64_t j = 0; j < vm.numSegments; ++j) {
            const uint64_t n = vm.segmentNumTriangles[j];
            if (n == 0) continue; // "Removed %d empty geometry segments" (scene.cpp:641-646)
            if (vm.normalUvBufferOffset + 24 * (base + n) > v.raw.size()) throw Error("vertex data beyond the end of " + path);
            s.qpos.emplace_back(3 * n);
            s.qnu.emplace_back(3 * n);
            std::memcpy(s.qpos.back().data(), v.raw.data() + vm.vertexBufferOffset + 24 * base, 24 * n);
            std::memcpy(s.qnu.back().data(), v.raw.data() + vm.normalUvBufferOffset + 24 * base, 24 * n);
            RptrGeometryDesc g;
            std::memset(&g, 0, sizeof(g));
            g.qpos = s.qpos.back().data();
            g.qnrm_uv = s.qnu.back().data();
            g.num_tris = (uint32_t)n;
            g.has_normals = g.has_uvs = 1;
            std::memcpy(g.quantized_scaling, vm.vertexScale, 12);
            std::memcpy(g.quantized_offset, vm.vertexOffset, 12);
            s.geometries.push_back(g);
            s.offsets[i].push_back(vm.segmentMaterialBaseOffsets[j]);
            base += n;
        }
        RptrMeshDesc md;
        md.first_geometry = first;
        md.num_geometries = (uint32_t)s.geometries.size() - first;
        md.dynamic = 0;
        s.meshes.push_back(md);
        RptrParameterizedMeshDesc pm;
        pm.mesh = (uint32_t)i;
        pm.tri_material_ids = nullptr;
        if (vm.numSegments == 1 && vm.numMaterialsInRange > 1) { // scene.cpp:654-658
            if (vm.materialIdBufferOffset + (uint64_t)vm.materialIdSize * vm.numTriangles > v.raw.size()) throw Error("material ids beyond the end of " + path);
            s.tri_ids[i].resize((size_t)vm.numTriangles);
            for (uint64_t t = 0; t < vm.numTriangles; ++t) // 16-bit ids: the backend keeps 8 bits per triangle (static_cast<uint8_t>, render_vulkan.cpp:1114-1126)
                s.tri_ids[i][(size_t)t] = v.raw[(size_t)(vm.materialIdBufferOffset + (uint64_t)vm.materialIdSize * t)];
            s.offsets[i].assign(1, vm.materialIdBufferBase);
            pm.tri_material_ids = s.tri_ids[i].data();
        }
        else if (lp.dynamic_meshes && !lp.ignore_animation && md.num_geometries > 0) {
            // scene.cpp:658-706: `_SHADERMESH_<name>` / `_SHADERSUBMESH_<name>` in a segment material's extended name makes the mesh
            // dynamic (the named vertex program is the host application's: rptr_hip_update_vertices)
            for (uint64_t j = 0; j < vm.numSegments; ++j) {
                const int32_t off = vm.segmentMaterialBaseOffsets[j];
                if (off < 0 || (size_t)off >= extended_names.size()) continue;
                const std::string &ext = extended_names[(size_t)off];
                if (ext.find("_SHADERMESH_") != std::string::npos || ext.find("_SHADERSUBMESH_") != std::string::npos)
                    s.meshes.back().dynamic |= lp.small_deformation ? RPTR_MESH_SUBTLY_DYNAMIC : RPTR_MESH_DYNAMIC;
            }
        }
        pm.material_offsets = s.offsets[i].data();
        s.pmeshes.push_back(pm);
    }
    for (size_t idx = 0; idx < v.instances.size(); ++idx) { // scene.cpp:722-749: only the base level of a LoD group is instanced
        const InstanceHeader &vi = v.instances[idx];
        const LodGroup &lod = v.lodGroups[(size_t)v.meshes[(size_t)vi.meshId].lodGroup];
        if (!lod.meshIds.empty() && lod.meshIds[0] != vi.meshId) continue;
        if (lp.instance_pruning_probability != 0.0f && halton2((uint32_t)idx) < lp.instance_pruning_probability) continue;
        const uint64_t at = transform_offset(vi.transformIndex, v.numStaticTransforms, v.numAnimatedTransforms, frame) * QUANTIZED_TRANSFORM_SIZE;
        if (at + QUANTIZED_TRANSFORM_SIZE > v.transforms.size()) throw Error("transform index beyond the table of " + path);
        RptrInstanceDesc in;
        instance_transform(v.transforms.data() + at, in.transform);
        in.parameterized_mesh = (uint32_t)vi.meshId;
        if (remove_first_lods > 0 && lod.meshIds.size() > 1) // SceneLoaderParams::PerFile::remove_first_LODs (scene.cpp:801-815, :229-246)
            in.parameterized_mesh = (uint32_t)lod.meshIds[std::min((size_t)remove_first_lods, lod.meshIds.size() - 1)];
        s.instances.push_back(in);
    }
    auto add_texture = [&](std::vector<uint8_t> rgba, uint32_t w, uint32_t h, bool srgb, uint32_t levels = 0) {
        s.texels.push_back(std::move(rgba));
        RptrTextureDesc t;
        t.rgba8 = nullptr; // (set below: the vector of vectors may still move)
        t.width = w;
        t.height = h;
        t.srgb = srgb ? 1u : 0u;
        t.mip_levels = levels;
        s.textures.push_back(t);
    };
    for (size_t i = 0; i < v.materialNames.size(); ++i) {
        const std::string &name = v.materialNames[i];
        const std::string &extended_name = extended_names[i];
        float emission = 0.0f, emitter_color[3] = {0, 0, 0}, transmission[4] = {0.0f, 1.5f, 0.0f, 0.0f};
        std::vector<float> em;
        if (read_params(tex_dir + name + "_EmissionIntensity.txt", 4, em)) {
            if (em.size() == 1) {
                std::vector<float> col;
                if (read_params(tex_dir + name + "_BaseColor.txt", 3, col) && !(col.empty() || col.size() == 3))
                    throw Error("Three color components expected for emission base color");
                col.resize(3, 0.0f);
                em.insert(em.end(), col.begin(), col.end());
            } else if (!(em.empty() || em.size() == 4))
                throw Error("One or four components expected for emission intensity + base color");
            if (!em.empty()) {
                emission = em[0];
                for (int k = 0; k < 3; ++k) emitter_color[k] = em[1 + (size_t)k];
            }
        }
        std::vector<float> tr;
        if (read_params(tex_dir + name + "_SpecularTransmission.txt", 4, tr))
            for (size_t k = 0; k < tr.size() && k < 4; ++k) transmission[k] = tr[k];
        RptrBaseMaterial mat = default_material();
        const uint32_t tid = 3u * (uint32_t)i;
        Texture col = ignore_textures ? Texture{} : read_vkt(tex_dir + name + "_BaseColor.vkt");
        bool has_alpha = false;
        if (col.present) {
            has_alpha = col.format == FMT_BC1_RGBA_UNORM || col.format == FMT_BC1_RGBA_SRGB || col.format == FMT_BC3_UNORM || col.format == FMT_BC3_SRGB ||
                        col.format == FMT_RGBA8_UNORM || col.format == FMT_RGBA8_SRGB;
            add_texture(std::move(col.rgba), (uint32_t)col.width, (uint32_t)col.height, true, col.levels);
        } else
            add_texture({255, 255, 255, 255}, 1, 1, true);
        if (!has_alpha) mat.flags |= RPTR_BASE_MATERIAL_NOALPHA;
        {
            const uint32_t bits = 0x80000000u | tid;
            std::memcpy(&mat.base_color[0], &bits, 4);
        }
        Texture nrm = ignore_textures ? Texture{} : read_vkt(tex_dir + name + "_Normal.vkt");
        if (nrm.present)
            add_texture(std::move(nrm.rgba), (uint32_t)nrm.width, (uint32_t)nrm.height, false, nrm.levels);
        else
            add_texture({127, 127, 127, 255}, 1, 1, false);
        mat.normal_map = (int32_t)tid + 1;
        Texture spec = ignore_textures ? Texture{} : read_vkt(tex_dir + name + "_Specular.vkt");
        if (spec.present)
            add_texture(std::move(spec.rgba), (uint32_t)spec.width, (uint32_t)spec.height, false, spec.levels);
        else
            add_texture({255, 127, 0, 255}, 1, 1, false);
        mat.roughness = textured_param(tid + 2, 1);
        mat.metallic = textured_param(tid + 2, 2);
        if (load_specularity) mat.specular = textured_param(tid + 2, 0);
        if (emission > 0.0f) {
            if (emitter_color[0] != 0.0f || emitter_color[1] != 0.0f || emitter_color[2] != 0.0f) std::memcpy(mat.base_color, emitter_color, 12);
            mat.emission_intensity = emission;
        }
        mat.specular_transmission = transmission[0];
        if (transmission[0] != 0.0f && extended_name.find("twosided") == std::string::npos && extended_name.find("doublesided") == std::string::npos &&
            extended_name.find("TwoSided") == std::string::npos && extended_name.find("DoubleSided") == std::string::npos)
            mat.flags |= RPTR_BASE_MATERIAL_ONESIDED;
        mat.ior = transmission[1];
        s.materials.push_back(mat);
    }
    for (size_t i = 0; i < s.textures.size(); ++i) s.textures[i].rgba8 = s.texels[i].data();
    // application defaults (what scenes.py gives a scene read from a file): RenderParams / LightSamplingConfig defaults, the "default"
    // sky of the package data, a camera that sees the whole scene
    s.render_params = RptrRenderParams{1, RPTR_MAX_PATH_DEPTH, RPTR_DEFAULT_RR_PATH_DEPTH, 0, 0.f, 2.5f, 1.f, 4.f, 0, 0, 0.f, -1, 0, 8, 0, 1, 35.f, 0, 0, 0};
    s.lighting = RptrLightSamplingConfig{0.f, 16, 15.f, 0.f};
    lights::prepare_lights(s);
    s.scene_params = load_sky_params(data_dir + "/sky_params.json", "default", !s.lights.empty());
    s.scene_params.normal_z_scale = 1.0f; // 1 / bump_scale
    { // the default camera of vks.py: _default_camera (bounds of up to ~4096 vertices per geometry and instance, in double)
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (const RptrInstanceDesc &in : s.instances) {
            const RptrMeshDesc &mesh = s.meshes[s.pmeshes[in.parameterized_mesh].mesh];
            for (uint32_t gi = mesh.first_geometry; gi < mesh.first_geometry + mesh.num_geometries; ++gi) {
                const RptrGeometryDesc &g = s.geometries[gi];
                const size_t nv = (size_t)3 * g.num_tris, step = nv / 4096 > 1 ? nv / 4096 : 1;
                for (size_t k = 0; k < nv; k += step) {
                    const lights::V3 p = lights::dequantize_position(g.qpos[k], g.quantized_scaling, g.quantized_offset);
                    for (int r = 0; r < 3; ++r) {
                        const double w = (double(p.x) * double(in.transform[4 * r + 0]) + double(p.y) * double(in.transform[4 * r + 1]) +
                                          double(p.z) * double(in.transform[4 * r + 2])) + double(in.transform[4 * r + 3]);
                        lo[r] = std::min(lo[r], w);
                        hi[r] = std::max(hi[r], w);
                    }
                }
            }
        }
        if (!(std::isfinite(lo[0]) && std::isfinite(lo[1]) && std::isfinite(lo[2]))) {
            for (int r = 0; r < 3; ++r) {
                lo[r] = 0.0;
                hi[r] = 1.0;
            }
        }
        double c[3], d2 = 0.0;
        for (int r = 0; r < 3; ++r) {
            c[r] = 0.5 * (lo[r] + hi[r]);
            d2 += (hi[r] - lo[r]) * (hi[r] - lo[r]);
        }
        const double rad = 0.5 * std::sqrt(d2);
        const float eye[3] = {float(c[0]), float(c[1] + 0.35 * rad), float(c[2] + 1.6 * rad)}, center[3] = {float(c[0]), float(c[1]), float(c[2])};
        float dir[3] = {center[0] - eye[0], center[1] - eye[1], center[2] - eye[2]};
        const float inv = 1.0f / std::sqrt((dir[0] * dir[0] + dir[1] * dir[1]) + dir[2] * dir[2]);
        for (int r = 0; r < 3; ++r) {
            s.camera.pos[r] = eye[r];
            s.camera.dir[r] = dir[r] * inv;
        }
        s.camera.up[0] = 0.f;
        s.camera.up[1] = 1.f;
        s.camera.up[2] = 0.f;
        s.camera.fovy = 50.0f;
    }
    return s;
}

} // namespace vks
} // namespace rptr

// read_image.hpp -- reads back what the validation / profiling / data-capture modes write (and what a build of the reference writes where
// no wavelet decoder is needed): single-part scan-line OpenEXR files with compression NONE, RLE, ZIPS or ZIP (zlib), channels of type HALF,
// FLOAT or UINT, every channel decoded to a float plane; and PFM. No tinyexr (the reference's util/compare_exr.cpp:16-49 loads through
// it with requested_pixel_types = FLOAT: the same result for these files). PIZ / PXR24 / B44 / DWA and tiled files are refused by name.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace rptr {

struct PlanarImage {
    int width = 0, height = 0;
    std::vector<std::string> names;        // EXR: the file's channel list (alphabetical); PFM: "R" "G" "B" or "Y"
    std::vector<std::vector<float>> plane; // one width * height plane per channel, top row first
};

namespace detail {
inline std::vector<uint8_t> slurp(const std::string &path) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    std::vector<uint8_t> b;
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) b.insert(b.end(), buf, buf + n);
    std::fclose(f);
    return b;
}
inline float half_bits_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    uint32_t u;
    if (e == 0) {
        if (m == 0) u = sign;
        else {
            int shift = 0;
            uint32_t mm = m;
            while (!(mm & 0x400u)) mm <<= 1, ++shift;
            u = sign | ((uint32_t)(113 - shift) << 23) | ((mm & 0x3FFu) << 13);
        }
    } else if (e == 31) u = sign | 0x7F800000u | (m << 13);
    else u = sign | ((e + 112u) << 23) | (m << 13);
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
// the byte shuffle + delta predictor OpenEXR's ZIP / ZIPS / RLE compression applies before the entropy coder, undone; rle: run-length
// coded (count >= 0: count + 1 copies of the next byte, count < 0: -count literal bytes) instead of deflated
inline void exr_unzip(const uint8_t *src, size_t src_size, uint8_t *dst, size_t dst_size, bool rle = false) {
    std::vector<uint8_t> t(dst_size);
    if (rle) {
        size_t in = 0, out = 0;
        while (in < src_size) {
            const int count = (int8_t)src[in++];
            const size_t n = count < 0 ? (size_t)(-count) : (size_t)count + 1;
            if (out + n > dst_size || in + (count < 0 ? n : 1) > src_size) throw std::runtime_error("corrupt RLE block in EXR file");
            if (count < 0) {
                std::memcpy(t.data() + out, src + in, n);
                in += n;
            } else
                std::memset(t.data() + out, src[in++], n);
            out += n;
        }
        if (out != dst_size) throw std::runtime_error("corrupt RLE block in EXR file");
    } else {
        uLongf got = (uLongf)dst_size;
        if (uncompress(t.data(), &got, src, (uLong)src_size) != Z_OK || got != dst_size) throw std::runtime_error("corrupt ZIP block in EXR file");
    }
    for (size_t i = 1; i < dst_size; ++i) t[i] = (uint8_t)(t[i - 1] + t[i] - 128);
    const size_t half = (dst_size + 1) / 2;
    for (size_t i = 0; i < dst_size; ++i) dst[i] = (i & 1) ? t[half + i / 2] : t[i / 2];
}
} // namespace detail

inline PlanarImage read_exr(const std::string &path) {
    using namespace detail;
    const std::vector<uint8_t> raw = slurp(path);
    auto u32 = [&](size_t at) {
        if (at + 4 > raw.size()) throw std::runtime_error("truncated EXR file " + path);
        uint32_t v;
        std::memcpy(&v, raw.data() + at, 4);
        return v;
    };
    if (u32(0) != 20000630u) throw std::runtime_error(path + " is not an OpenEXR file");
    const uint32_t version = u32(4);
    if ((version & 0xFFu) != 2u || (version & 0x1A00u)) throw std::runtime_error("Tiled images are not supported."); // tiled / multi-part / deep
    std::map<std::string, std::vector<uint8_t>> attrs;
    size_t at = 8;
    while (at < raw.size() && raw[at] != 0) {
        const size_t name_end = std::find(raw.begin() + at, raw.end(), 0) - raw.begin();
        const size_t type_end = std::find(raw.begin() + name_end + 1, raw.end(), 0) - raw.begin();
        const uint32_t size = u32(type_end + 1);
        if (type_end + 5 + size > raw.size()) throw std::runtime_error("truncated EXR header in " + path);
        attrs[std::string(raw.begin() + at, raw.begin() + name_end)] = std::vector<uint8_t>(raw.begin() + type_end + 5, raw.begin() + type_end + 5 + size);
        at = type_end + 5 + size;
    }
    ++at;
    for (const char *need : {"channels", "compression", "dataWindow", "lineOrder"})
        if (!attrs.count(need)) throw std::runtime_error(std::string("EXR header without ") + need + " in " + path);
    const int compression = attrs["compression"][0];
    static const char *comp_names[] = {"NONE", "RLE", "ZIPS", "ZIP", "PIZ", "PXR24", "B44", "B44A", "DWAA", "DWAB"};
    if (!(compression >= 0 && compression <= 3))
        throw std::runtime_error(path + ": compression " + (compression < 10 ? comp_names[compression] : "?") + " is not supported (NONE, RLE, ZIPS, ZIP are)");
    int32_t box[4];
    std::memcpy(box, attrs["dataWindow"].data(), 16);
    PlanarImage img;
    img.width = box[2] - box[0] + 1;
    img.height = box[3] - box[1] + 1;
    if (img.width <= 0 || img.height <= 0) throw std::runtime_error("empty data window in " + path);
    std::vector<int> types;
    {
        const std::vector<uint8_t> &ch = attrs["channels"];
        size_t k = 0;
        while (k < ch.size() && ch[k] != 0) {
            const size_t e = std::find(ch.begin() + k, ch.end(), 0) - ch.begin();
            if (e + 17 > ch.size()) throw std::runtime_error("bad channel list in " + path);
            img.names.emplace_back(ch.begin() + k, ch.begin() + e);
            int32_t t, sx, sy;
            std::memcpy(&t, ch.data() + e + 1, 4);
            std::memcpy(&sx, ch.data() + e + 9, 4);
            std::memcpy(&sy, ch.data() + e + 13, 4);
            if (sx != 1 || sy != 1 || t < 0 || t > 2) throw std::runtime_error("subsampled or unknown channel type in " + path);
            types.push_back(t);
            k = e + 17;
        }
    }
    const size_t nch = img.names.size();
    size_t row_bytes = 0;
    for (int t : types) row_bytes += (size_t)img.width * (t == 1 ? 2 : 4);
    img.plane.assign(nch, std::vector<float>((size_t)img.width * img.height));
    const int lines_per_block = compression == 3 ? 16 : 1;
    const int blocks = (img.height + lines_per_block - 1) / lines_per_block;
    std::vector<uint8_t> block;
    for (int b = 0; b < blocks; ++b) {
        uint64_t off;
        if (at + 8ull * (b + 1) > raw.size()) throw std::runtime_error("truncated offset table in " + path);
        std::memcpy(&off, raw.data() + at + 8ull * b, 8);
        const int32_t y0 = (int32_t)u32(off) - box[1];
        const uint32_t size = u32(off + 4);
        if (off + 8 + size > raw.size() || y0 < 0 || y0 >= img.height) throw std::runtime_error("bad scan-line block in " + path);
        const int lines = std::min(lines_per_block, img.height - y0);
        const size_t want = row_bytes * (size_t)lines;
        block.resize(want);
        if (compression == 0 || size == want) {
            if (size != want) throw std::runtime_error("scan-line block of the wrong size in " + path);
            std::memcpy(block.data(), raw.data() + off + 8, want);
        } else
            exr_unzip(raw.data() + off + 8, size, block.data(), want, compression == 1);
        const uint8_t *p = block.data();
        for (int l = 0; l < lines; ++l)
            for (size_t c = 0; c < nch; ++c) {
                float *dst = img.plane[c].data() + (size_t)(y0 + l) * img.width;
                for (int x = 0; x < img.width; ++x) {
                    if (types[c] == 1) {
                        uint16_t h;
                        std::memcpy(&h, p, 2);
                        dst[x] = half_bits_to_float(h);
                        p += 2;
                    } else if (types[c] == 2) {
                        std::memcpy(dst + x, p, 4);
                        p += 4;
                    } else {
                        uint32_t u;
                        std::memcpy(&u, p, 4);
                        dst[x] = (float)u;
                        p += 4;
                    }
                }
            }
    }
    return img;
}

// "PF" (3 channels) / "Pf" (1 channel), negative scale = little endian, bottom row first (util/write_image.cpp:34-66 writes this)
inline PlanarImage read_pfm(const std::string &path) {
    const std::vector<uint8_t> raw = detail::slurp(path);
    int w = 0, h = 0, used = 0;
    char kind = 0;
    float scale = 0.f;
    if (std::sscanf(reinterpret_cast<const char *>(raw.data()), "P%c %d %d %f%n", &kind, &w, &h, &scale, &used) != 4 || (kind != 'F' && kind != 'f') || w <= 0 || h <= 0)
        throw std::runtime_error(path + " is not a PFM file");
    if (scale >= 0.f) throw std::runtime_error(path + ": big-endian PFM files are not supported");
    const size_t nch = kind == 'F' ? 3 : 1, at = (size_t)used + 1;
    if (at + (size_t)w * h * nch * 4 > raw.size()) throw std::runtime_error("truncated PFM file " + path);
    PlanarImage img;
    img.width = w;
    img.height = h;
    img.names = nch == 3 ? std::vector<std::string>{"R", "G", "B"} : std::vector<std::string>{"Y"};
    img.plane.assign(nch, std::vector<float>((size_t)w * h));
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            for (size_t c = 0; c < nch; ++c)
                std::memcpy(&img.plane[c][(size_t)(h - 1 - y) * w + x], raw.data() + at + (((size_t)y * w + x) * nch + c) * 4, 4);
    return img;
}

inline PlanarImage read_image(const std::string &path) {
    const size_t n = path.size();
    if (n >= 4 && path.compare(n - 4, 4, ".pfm") == 0) return read_pfm(path);
    return read_exr(path);
}

} // namespace rptr

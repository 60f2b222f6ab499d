// write_image.hpp -- the three image outputs of the reference's harness (util/write_image.cpp:13-190, selected by --exr (default),
// --pfm, --png: cmdline.cpp:450-460), written without the reference's stb / tinyexr dependencies:
//   * PFM  : "PF\n<w> <h>\n-1.0\n" + RGB float rows, bottom row first -- the reference's own layout (write_image.cpp:34-64)
//   * EXR  : single-part scan-line OpenEXR, channels A, B, G, R (alphabetical, as write_image.cpp:100-128 orders them), FLOAT for
//            the accumulation buffer, HALF for the AOV images, no compression (the reference asks tinyexr for PIZ in validation
//            mode and NONE in profiling / data-capture mode: a storage detail, any OpenEXR reader returns the same pixels)
//   * PNG  : 8-bit RGBA, filter 0, zlib stream of stored (uncompressed) deflate blocks
// Header-only, C++17, no dependencies.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace rptr {

inline bool write_pfm(const std::string &prefix, unsigned width, unsigned height, unsigned channels, const float *pixels) {
    if (width == 0 || height == 0 || channels < 3 || !pixels) return false;
    const std::string path = prefix + ".pfm";
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    std::fprintf(f, "PF\n%i %i\n-1.0\n", width, height);
    std::vector<float> rgb((size_t)width * height * 3);
    for (unsigned y = 0; y < height; ++y) // the file stores the bottom row first
        for (unsigned x = 0; x < width; ++x)
            for (unsigned j = 0; j < 3; ++j) rgb[((size_t)width * (height - y - 1) + x) * 3 + j] = pixels[((size_t)width * y + x) * channels + j];
    const bool ok = std::fwrite(rgb.data(), sizeof(float), rgb.size(), f) == rgb.size();
    std::fclose(f);
    return ok;
}

namespace detail {
inline void put_u32le(std::vector<uint8_t> &o, uint32_t v) {
    for (int k = 0; k < 4; ++k) o.push_back(uint8_t(v >> (8 * k)));
}
inline void put_u64le(std::vector<uint8_t> &o, uint64_t v) {
    for (int k = 0; k < 8; ++k) o.push_back(uint8_t(v >> (8 * k)));
}
inline void put_u32be(std::vector<uint8_t> &o, uint32_t v) {
    for (int k = 3; k >= 0; --k) o.push_back(uint8_t(v >> (8 * k)));
}
inline void put_str(std::vector<uint8_t> &o, const char *s) { // null-terminated
    while (*s) o.push_back(uint8_t(*s++));
    o.push_back(0);
}
inline void put_attr(std::vector<uint8_t> &o, const char *name, const char *type, const std::vector<uint8_t> &value) {
    put_str(o, name);
    put_str(o, type);
    put_u32le(o, (uint32_t)value.size());
    o.insert(o.end(), value.begin(), value.end());
}
inline uint32_t crc32(const uint8_t *p, size_t n, uint32_t crc = 0) {
    static uint32_t table[256];
    static bool have = false;
    if (!have) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        have = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    return ~crc;
}
} // namespace detail

// T = float (PIXELTYPE FLOAT = 2) or uint16_t holding IEEE halfs (PIXELTYPE HALF = 1); pixels are RGBA interleaved, top row first
template <class T>
inline bool write_exr(const std::string &prefix, unsigned width, unsigned height, unsigned channels, const T *pixels) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 2, "float or half");
    if (width == 0 || height == 0 || channels != 4 || !pixels) return false;
    using namespace detail;
    std::vector<uint8_t> o;
    put_u32le(o, 20000630u); // magic
    put_u32le(o, 2u);        // version 2, single-part scan lines
    {
        std::vector<uint8_t> ch;
        const char *names[4] = {"A", "B", "G", "R"};
        for (int c = 0; c < 4; ++c) {
            put_str(ch, names[c]);
            put_u32le(ch, sizeof(T) == 4 ? 2u : 1u); // pixel type
            ch.push_back(0);                          // pLinear
            ch.push_back(0), ch.push_back(0), ch.push_back(0);
            put_u32le(ch, 1u), put_u32le(ch, 1u);     // sampling
        }
        ch.push_back(0);
        put_attr(o, "channels", "chlist", ch);
    }
    put_attr(o, "compression", "compression", {0});
    {
        std::vector<uint8_t> box;
        put_u32le(box, 0u), put_u32le(box, 0u), put_u32le(box, width - 1), put_u32le(box, height - 1);
        put_attr(o, "dataWindow", "box2i", box);
        put_attr(o, "displayWindow", "box2i", box);
    }
    put_attr(o, "lineOrder", "lineOrder", {0}); // increasing y
    {
        std::vector<uint8_t> v;
        const float one = 1.0f, zero = 0.0f;
        uint32_t b;
        std::memcpy(&b, &one, 4);
        put_u32le(v, b);
        put_attr(o, "pixelAspectRatio", "float", v);
        put_attr(o, "screenWindowWidth", "float", v);
        v.clear();
        std::memcpy(&b, &zero, 4);
        put_u32le(v, b), put_u32le(v, b);
        put_attr(o, "screenWindowCenter", "v2f", v);
    }
    o.push_back(0); // end of header
    const size_t row_bytes = (size_t)width * 4 * sizeof(T);
    const size_t table_at = o.size();
    const uint64_t first = table_at + 8ull * height;
    for (unsigned y = 0; y < height; ++y) put_u64le(o, first + (uint64_t)y * (8 + row_bytes));
    std::vector<T> row((size_t)width * 4);
    const int src_of[4] = {3, 2, 1, 0}; // A, B, G, R from RGBA
    for (unsigned y = 0; y < height; ++y) {
        put_u32le(o, y);
        put_u32le(o, (uint32_t)row_bytes);
        for (int c = 0; c < 4; ++c)
            for (unsigned x = 0; x < width; ++x) row[(size_t)c * width + x] = pixels[((size_t)y * width + x) * 4 + src_of[c]];
        const uint8_t *rb = reinterpret_cast<const uint8_t *>(row.data());
        o.insert(o.end(), rb, rb + row_bytes);
    }
    FILE *f = std::fopen((prefix + ".exr").c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(o.data(), 1, o.size(), f) == o.size();
    std::fclose(f);
    return ok;
}

inline bool write_png(const std::string &prefix, unsigned width, unsigned height, unsigned channels, const unsigned char *pixels) {
    if (width == 0 || height == 0 || channels != 4 || !pixels) return false;
    using namespace detail;
    // raw scan lines with filter type 0
    std::vector<uint8_t> raw;
    raw.reserve((size_t)height * (1 + (size_t)width * 4));
    for (unsigned y = 0; y < height; ++y) {
        raw.push_back(0);
        raw.insert(raw.end(), pixels + (size_t)y * width * 4, pixels + (size_t)(y + 1) * width * 4);
    }
    // zlib: header, stored blocks of at most 65535 bytes, Adler-32
    std::vector<uint8_t> z;
    z.push_back(0x78), z.push_back(0x01);
    uint32_t a = 1, b = 0;
    for (size_t at = 0; at < raw.size() || at == 0;) {
        const size_t n = raw.size() - at < 65535 ? raw.size() - at : 65535;
        z.push_back(at + n == raw.size() ? 1 : 0);
        z.push_back(uint8_t(n)), z.push_back(uint8_t(n >> 8)), z.push_back(uint8_t(~n)), z.push_back(uint8_t((~n) >> 8));
        z.insert(z.end(), raw.begin() + at, raw.begin() + at + n);
        for (size_t i = at; i < at + n; ++i) {
            a = (a + raw[i]) % 65521u;
            b = (b + a) % 65521u;
        }
        at += n;
        if (n == 0) break;
    }
    put_u32be(z, (b << 16) | a);
    std::vector<uint8_t> o = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    auto chunk = [&](const char type[4], const std::vector<uint8_t> &data) {
        put_u32be(o, (uint32_t)data.size());
        std::vector<uint8_t> td(type, type + 4);
        td.insert(td.end(), data.begin(), data.end());
        o.insert(o.end(), td.begin(), td.end());
        put_u32be(o, crc32(td.data(), td.size()));
    };
    std::vector<uint8_t> ihdr;
    put_u32be(ihdr, width), put_u32be(ihdr, height);
    ihdr.push_back(8), ihdr.push_back(6), ihdr.push_back(0), ihdr.push_back(0), ihdr.push_back(0); // 8 bit, RGBA
    chunk("IHDR", ihdr);
    chunk("IDAT", z);
    chunk("IEND", {});
    FILE *f = std::fopen((prefix + ".png").c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(o.data(), 1, o.size(), f) == o.size();
    std::fclose(f);
    return ok;
}

} // namespace rptr

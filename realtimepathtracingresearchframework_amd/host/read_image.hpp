// read_image.hpp -- reads back what the validation / profiling / data-capture modes write and what a build of the reference writes:
// single-part scan-line OpenEXR files with compression NONE, RLE, ZIPS, ZIP (zlib) or PIZ -- the reference's --validation images are PIZ
// (libapp/app_state.cpp:476, util/write_image.cpp:150-151) --, channels of type HALF, FLOAT or UINT, every channel decoded to a float
// plane; and PFM. No tinyexr (the reference's util/compare_exr.cpp:16-49 loads through it with requested_pixel_types = FLOAT: the same
// result for these files). PXR24 / B44 / DWA and tiled files are refused by name.
// PIZ (OpenEXR technical introduction, "PIZ"; ImfPizCompressor / ImfHuf / ImfWav of the OpenEXR library are the published statement):
// per block of 32 scan lines the 16-bit words of all channels (a FLOAT / UINT sample is two words) -> a bitmap of the values that occur
// and a look-up table that packs them densely -> a 2D Haar-like wavelet per channel (14-bit variant when the packed range allows, 16-bit
// modular arithmetic otherwise) -> canonical Huffman code with a run-length symbol. Decoding runs the three steps backwards.
#pragma once
#include <zlib.h>

#include <algorithm>

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
// ---- PIZ: MSB-first bit reader over a byte range
struct BitReader {
    const uint8_t *p, *end;
    uint64_t acc = 0;
    int have = 0;
    uint64_t consumed = 0;
    uint32_t get(int n) { // n <= 32
        while (have < n) {
            acc = (acc << 8) | (p < end ? *p++ : 0u);
            have += 8;
        }
        have -= n;
        consumed += (uint64_t)n;
        return (uint32_t)((acc >> have) & ((n == 32) ? 0xFFFFFFFFull : ((1ull << n) - 1ull)));
    }
};
// the Huffman stage: 20-byte header (im, iM, table bytes, data bits, reserved), the code lengths of symbols im .. iM packed in 6 bits each
// (59 .. 62: a run of 2 .. 5 zero lengths, 63: 8 more bits = run - 6), canonical codes (longest codes get the smallest values), symbol iM
// = "repeat the previous word": 8 bits of count follow
inline void piz_huf_uncompress(const uint8_t *src, size_t n_src, uint16_t *out, size_t n_out) {
    if (n_src == 0) {
        if (n_out) throw std::runtime_error("corrupt PIZ block (no Huffman data)");
        return;
    }
    if (n_src < 20) throw std::runtime_error("corrupt PIZ block (Huffman header)");
    uint32_t hdr[5];
    std::memcpy(hdr, src, 20);
    const uint32_t im = hdr[0], iM = hdr[1], table_bytes = hdr[2], n_bits = hdr[3];
    const uint32_t ENC = (1u << 16) + 1u;
    if (im >= ENC || iM >= ENC || im > iM || 20ull + table_bytes > n_src) throw std::runtime_error("corrupt PIZ block (Huffman table range)");
    std::vector<uint8_t> len(ENC, 0);
    {
        BitReader br{src + 20, src + 20 + table_bytes};
        for (uint32_t s = im; s <= iM;) {
            const uint32_t l = br.get(6);
            if (l == 63u) {
                const uint32_t run = br.get(8) + 6u;
                if (s + run > iM + 1u) throw std::runtime_error("corrupt PIZ block (Huffman table run)");
                s += run;
            } else if (l >= 59u) {
                const uint32_t run = l - 59u + 2u;
                if (s + run > iM + 1u) throw std::runtime_error("corrupt PIZ block (Huffman table run)");
                s += run;
            } else
                len[s++] = (uint8_t)l;
        }
    }
    // canonical codes: count per length; base of the longest length is 0, every shorter length starts at (base + count) / 2 of the next longer
    uint64_t count[59] = {0}, base[59] = {0};
    for (uint32_t s = im; s <= iM; ++s) count[len[s]]++;
    count[0] = 0;
    {
        uint64_t c = 0;
        for (int l = 58; l >= 1; --l) {
            const uint64_t nc = (c + count[l]) >> 1;
            base[l] = c;
            c = nc;
        }
    }
    std::vector<uint32_t> first(60, 0), syms;
    syms.reserve(iM - im + 1);
    for (int l = 1; l <= 58; ++l) {
        first[l] = (uint32_t)syms.size();
        for (uint32_t s = im; s <= iM; ++s)
            if (len[s] == l) syms.push_back(s);
    }
    BitReader br{src + 20 + table_bytes, src + n_src};
    size_t o = 0;
    while (o < n_out) {
        uint64_t code = 0;
        int l = 0;
        uint32_t sym = ENC;
        while (l < 58) {
            code = (code << 1) | br.get(1);
            ++l;
            if (count[l] && code >= base[l] && code - base[l] < count[l]) {
                sym = syms[first[l] + (uint32_t)(code - base[l])];
                break;
            }
        }
        if (sym == ENC || br.consumed > (uint64_t)n_bits + 7u) throw std::runtime_error("corrupt PIZ block (Huffman code)");
        if (sym == iM) { // run: the previous word `count` more times
            const uint32_t run = br.get(8);
            if (o == 0 || o + run > n_out) throw std::runtime_error("corrupt PIZ block (Huffman run)");
            for (uint32_t k = 0; k < run; ++k, ++o) out[o] = out[o - 1];
        } else
            out[o++] = (uint16_t)sym;
    }
}
// the wavelet stage backwards, in place: words at in[y * oy + x * ox]
inline void piz_wav2_decode(uint16_t *in, int nx, int ox, int ny, int oy, uint16_t max_value) {
    const bool w14 = max_value < (1u << 14);
    auto dec = [&](uint16_t l, uint16_t h, uint16_t &a, uint16_t &b) {
        if (w14) {
            const int ls = (int16_t)l, hs = (int16_t)h;
            const int ai = ls + (hs & 1) + (hs >> 1);
            a = (uint16_t)(int16_t)ai;
            b = (uint16_t)(int16_t)(ai - hs);
        } else {
            const int m = l, d = h;
            const int bb = (m - (d >> 1)) & 0xFFFF;
            const int aa = (d + bb - 0x8000) & 0xFFFF;
            b = (uint16_t)bb;
            a = (uint16_t)aa;
        }
    };
    const int n = nx > ny ? ny : nx;
    int p = 1;
    while (p <= n) p <<= 1;
    p >>= 1;
    int p2 = p;
    p >>= 1;
    while (p >= 1) {
        uint16_t *py = in;
        uint16_t *const ey = in + (ptrdiff_t)oy * (ny - p2);
        const ptrdiff_t oy1 = (ptrdiff_t)oy * p, oy2 = (ptrdiff_t)oy * p2, ox1 = (ptrdiff_t)ox * p, ox2 = (ptrdiff_t)ox * p2;
        uint16_t i00, i01, i10, i11;
        for (; py <= ey; py += oy2) {
            uint16_t *px = py;
            uint16_t *const ex = py + (ptrdiff_t)ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t *p01 = px + ox1, *p10 = px + oy1, *p11 = p10 + ox1;
                dec(*px, *p10, i00, i10);
                dec(*p01, *p11, i01, i11);
                dec(i00, i01, *px, *p01);
                dec(i10, i11, *p10, *p11);
            }
            if (nx & p) {
                uint16_t *p10 = px + oy1;
                dec(*px, *p10, i00, *p10);
                *px = i00;
            }
        }
        if (ny & p) {
            uint16_t *px = py;
            uint16_t *const ex = py + (ptrdiff_t)ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t *p01 = px + ox1;
                dec(*px, *p01, i00, *p01);
                *px = i00;
            }
        }
        p2 = p;
        p >>= 1;
    }
}
// one block of `lines` scan lines: `words_per_row[c]` 16-bit words per row of channel c; dst: the block in the file's uncompressed layout
inline void piz_decode_block(const uint8_t *src, size_t n_src, uint8_t *dst, size_t dst_size, const std::vector<int> &words_per_pixel, int width, int lines) {
    if (n_src < 4) throw std::runtime_error("corrupt PIZ block");
    uint16_t min_nz, max_nz;
    std::memcpy(&min_nz, src, 2);
    std::memcpy(&max_nz, src + 2, 2);
    std::vector<uint8_t> bitmap(8192, 0);
    size_t at = 4;
    if (min_nz <= max_nz) {
        const size_t nb = (size_t)max_nz - min_nz + 1;
        if (max_nz >= 8192 || at + nb > n_src) throw std::runtime_error("corrupt PIZ block (bitmap)");
        std::memcpy(bitmap.data() + min_nz, src + at, nb);
        at += nb;
    }
    std::vector<uint16_t> lut(65536, 0);
    uint32_t k = 0;
    for (uint32_t i = 0; i < 65536; ++i)
        if (i == 0 || (bitmap[i >> 3] & (1u << (i & 7)))) lut[k++] = (uint16_t)i;
    const uint16_t max_value = (uint16_t)(k - 1);
    if (at + 4 > n_src) throw std::runtime_error("corrupt PIZ block (length)");
    int32_t length;
    std::memcpy(&length, src + at, 4);
    at += 4;
    if (length < 0 || at + (size_t)length > n_src) throw std::runtime_error("corrupt PIZ block (length)");
    size_t n_words = 0;
    for (int w : words_per_pixel) n_words += (size_t)w * width * lines;
    if (n_words * 2 != dst_size) throw std::runtime_error("PIZ block of the wrong size");
    std::vector<uint16_t> tmp(n_words);
    piz_huf_uncompress(src + at, (size_t)length, tmp.data(), n_words);
    size_t start = 0;
    std::vector<size_t> starts;
    for (int w : words_per_pixel) {
        starts.push_back(start);
        for (int j = 0; j < w; ++j) piz_wav2_decode(tmp.data() + start + j, width, w, lines, width * w, max_value);
        start += (size_t)w * width * lines;
    }
    for (uint16_t &v : tmp) v = lut[v];
    // back to scan lines: row by row, channel by channel
    uint8_t *o = dst;
    for (int y = 0; y < lines; ++y)
        for (size_t c = 0; c < words_per_pixel.size(); ++c) {
            const size_t n = (size_t)words_per_pixel[c] * width;
            std::memcpy(o, tmp.data() + starts[c] + (size_t)y * n, n * 2);
            o += n * 2;
        }
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
    if (!(compression >= 0 && compression <= 4))
        throw std::runtime_error(path + ": compression " + (compression < 10 ? comp_names[compression] : "?") + " is not supported (NONE, RLE, ZIPS, ZIP, PIZ are)");
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
    const int lines_per_block = compression == 4 ? 32 : compression == 3 ? 16 : 1;
    std::vector<int> words_per_pixel;
    for (int t : types) words_per_pixel.push_back(t == 1 ? 1 : 2);
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
        } else if (compression == 4)
            piz_decode_block(raw.data() + off + 8, size, block.data(), want, words_per_pixel, img.width, lines);
        else
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

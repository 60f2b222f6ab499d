// rptr_compare REF CMP [CMP...] -- the second half of the reference's image regression harness (util/compare_exr.cpp; the first half is
// `--validation <prefix>`): every CMP image is compared with REF channel by channel, the relative error of a value is |ref - cmp| / ref
// (|cmp| where ref is 0, :71-76), an image with an error above 1e-6 anywhere "isn't the same" (:80-81), the error of every value is
// written to <CMP>_err.exr (32-bit float channels, :86-92), the exit code is -1 when any image differs or cannot be read (:99-130).
// Files: what host/read_image.hpp reads (scan-line EXR with NONE / RLE / ZIP / ZIPS compression, PFM).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

#include "read_image.hpp"
#include "write_image.hpp"

static bool compare(const rptr::PlanarImage &ref, const rptr::PlanarImage &cmp, const std::string &err_prefix) {
    if (ref.width != cmp.width || ref.height != cmp.height || ref.names.size() != cmp.names.size()) {
        std::fprintf(stderr, "Images must have the same size as the reference image\n");
        return false;
    }
    const size_t n = (size_t)ref.width * ref.height, nch = ref.names.size();
    bool equal = true;
    std::vector<std::vector<float>> err(nch, std::vector<float>(n));
    for (size_t c = 0; c < nch; ++c)
        for (size_t p = 0; p < n; ++p) {
            const float vref = ref.plane[c][p], vcmp = cmp.plane[c][p];
            const float rel = vref == 0.f ? std::fabs(vcmp) : std::fabs(vref - vcmp) / vref;
            err[c][p] = rel;
            if (rel > 1e-6f) equal = false;
        }
    // the error image: the channels in file order as R, G, B, A of an RGBA file (missing ones zero); write_exr adds ".exr"
    std::vector<float> rgba(n * 4, 0.f);
    auto slot = [&](const std::string &name, size_t index) {
        if (name == "R" || name == "Y") return 0;
        if (name == "G") return 1;
        if (name == "B") return 2;
        if (name == "A") return 3;
        return (int)std::min<size_t>(index, 3);
    };
    for (size_t c = 0; c < nch; ++c) {
        const int s = slot(ref.names[c], c);
        for (size_t p = 0; p < n; ++p) rgba[4 * p + s] = err[c][p];
    }
    if (!rptr::write_exr<float>(err_prefix, (unsigned)ref.width, (unsigned)ref.height, 4, rgba.data())) std::fprintf(stderr, "cannot write %s.exr\n", err_prefix.c_str());
    return equal;
}

int main(int argc, char **argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s REF CMP [CMP...]\n", argv[0]);
        return -1;
    }
    bool error = false;
    std::vector<rptr::PlanarImage> images;
    for (int i = 1; i < argc; ++i) {
        try {
            images.push_back(rptr::read_image(argv[i]));
        } catch (const std::exception &e) {
            std::fprintf(stderr, "%s\n", e.what());
            error = true;
        }
    }
    if (!error)
        for (size_t i = 1; i < images.size(); ++i) {
            std::printf("Comparing %s with %s\n", argv[1 + i], argv[1]);
            if (!compare(images[0], images[i], std::string(argv[1 + i]) + "_err")) {
                std::fprintf(stderr, "%s isn't the same as %s\n", argv[1 + i], argv[1]);
                error = true;
            }
        }
    return error ? -1 : 0;
}

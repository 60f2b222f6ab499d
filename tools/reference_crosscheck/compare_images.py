#!/usr/bin/env python3
"""compare_images.py REF.pfm CMP.pfm: RMSE over RGB (north_star's measure: < 1e-3 at fixed seed / spp), the share of pixels that differ by
more than 1e-3, and coverage -- pixels that are exactly the sky in one image and not in the other cannot be told from RGB alone, so coverage is
reported as the count of pixels where one image is NaN / Inf and the other is not. Exit code 1 when RMSE >= --tolerance."""
import sys

import numpy as np


def read_pfm(path):
    with open(path, "rb") as f:
        kind = f.readline().strip()
        w, h = (int(x) for x in f.readline().split())
        scale = float(f.readline())
        ch = 3 if kind == b"PF" else 1
        a = np.frombuffer(f.read(), "<f4" if scale < 0 else ">f4").reshape(h, w, ch)
    return a[::-1]   # (PFM rows run bottom-up)


def main():
    tol = 1e-3
    args = [a for a in sys.argv[1:] if not a.startswith("--tolerance")]
    for a in sys.argv[1:]:
        if a.startswith("--tolerance="):
            tol = float(a.split("=")[1])
    ref, cmp_ = read_pfm(args[0]), read_pfm(args[1])
    if ref.shape != cmp_.shape:
        print("sizes differ: %s vs %s" % (ref.shape, cmp_.shape))
        return 2
    fa, fb = np.isfinite(ref).all(axis=2), np.isfinite(cmp_).all(axis=2)
    both = fa & fb
    d = (ref.astype(np.float64) - cmp_.astype(np.float64))[both]
    rmse = float(np.sqrt(np.mean(d ** 2))) if d.size else 0.0
    off = int((np.abs(d).max(axis=1) > 1e-3).sum()) if d.size else 0
    print("%s vs %s: RMSE %.3g over %d pixels (tolerance %g), %d pixels differ by more than 1e-3 (max %.3g), %d pixels finite in one image only; means %.5f / %.5f"
          % (args[0], args[1], rmse, int(both.sum()), tol, off, float(np.abs(d).max()) if d.size else 0.0, int((fa != fb).sum()), float(ref[both].mean()), float(cmp_[both].mean())))
    return 0 if rmse < tol else 1


if __name__ == "__main__":
    sys.exit(main())

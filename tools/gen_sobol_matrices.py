#!/usr/bin/env python3
"""Package data `data/sobol_matrices_1024x32.u32`: the generator matrices of the first 1024 dimensions of the Sobol' sequence with the
direction numbers of S. Joe and F. Y. Kuo, "Constructing Sobol sequences with better two-dimensional projections", SIAM J. Sci. Comput.
30 (2008) -- the table new-joe-kuo-6.21201, read from the copy scipy ships (scipy/stats/_sobol_direction_numbers.npz: `poly` = the
primitive polynomials with both end coefficients, `vinit` = m_1..m_s).

Word j of dimension d is the direction number v_{j+1} = m_{j+1} * 2^(31-j); dimension 0 is the van der Corput sequence.  Recurrence
(Bratley & Fox, Algorithm 659): v_j = v_{j-s} ^ (v_{j-s} >> s) ^ XOR_{k=1..s-1} a_k v_{j-k}.

This is the table the reference keeps as rendering/pointsets/sobol_tables.h (SobolMatrix); tests/test_pointsets.py compares the two
word by word when /root/reference is present.  Run in the build container (needs scipy), output is committed.
"""
import os
import sys

import numpy as np


def sobol_matrices(n_dims=1024, bits=32):
    import scipy
    d = np.load(os.path.join(os.path.dirname(scipy.__file__), "stats", "_sobol_direction_numbers.npz"))
    poly, vinit = d["poly"], d["vinit"]
    out = np.zeros((n_dims, bits), dtype=np.uint64)
    out[0] = [1 << (bits - 1 - j) for j in range(bits)]
    for dim in range(1, n_dims):
        p = int(poly[dim])
        s = p.bit_length() - 1
        v = [0] * bits
        for j in range(min(s, bits)):
            v[j] = int(vinit[dim, j]) << (bits - 1 - j)
        for j in range(s, bits):
            x = v[j - s] ^ (v[j - s] >> s)
            for k in range(1, s):
                if (p >> (s - k)) & 1:
                    x ^= v[j - k]
            v[j] = x
        out[dim] = v
    return out.astype(np.uint32)


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "realtimepathtracingresearchframework_amd", "data", "sobol_matrices_1024x32.u32")
    m = sobol_matrices()
    m.astype("<u4").tofile(path)
    sys.stdout.write("%s: %d bytes\n" % (path, os.path.getsize(path)))

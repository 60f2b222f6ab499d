"""Writes the PIZ-compressed OpenEXR fixtures of tests/golden/piz/ (and the float planes they must decode to).

No OpenEXR / tinyexr encoder exists in this image (the reference's tinyexr submodule is absent, no Python binding is installed), so the
files are produced by THIS encoder: a second, independent statement of the published format (OpenEXR technical introduction; the library's
ImfPizCompressor / ImfHuf / ImfWav are the reference statement) in another language and shape than the C++ decoder it tests
(host/read_image.hpp): numpy arrays + heapq here, pointer walks there. What that pins: the two agree on bitmap / LUT, wavelet (both the
14-bit and the 16-bit variant, odd widths and heights, a short last block), canonical code assignment, zero-run table packing and the
run-length symbol. What it cannot pin: a shared misreading of the specification -- for that a file written by the reference's own writer
(util/write_image.cpp through tinyexr) is needed; drop one into tests/golden/piz/ as <name>.exr + <name>.f32 and the test picks it up.

python tests/golden/gen_piz_fixture.py  -> tests/golden/piz/*.exr, *.f32
"""
import heapq
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "piz")


# ---- wavelet (forward), words addressed as buf[y * oy + x * ox]
def wenc14(a, b):
    a_s, b_s = np.int16(np.uint16(a)), np.int16(np.uint16(b))
    ms = (int(a_s) + int(b_s)) >> 1
    ds = int(a_s) - int(b_s)
    return ms & 0xFFFF, ds & 0xFFFF


def wenc16(a, b):
    ao = (a + 0x8000) & 0xFFFF
    m = (ao + b) >> 1
    d = ao - b
    if d < 0:
        m = (m + 0x8000) & 0xFFFF
    return m & 0xFFFF, d & 0xFFFF


def wav2_encode(buf, start, nx, ox, ny, oy, max_value):
    enc = wenc14 if max_value < (1 << 14) else wenc16
    n = min(nx, ny)
    p, p2 = 1, 2
    while p2 <= n:
        oy1, oy2, ox1, ox2 = oy * p, oy * p2, ox * p, ox * p2
        py, ey = start, start + oy * (ny - p2)
        while py <= ey:
            px, ex = py, py + ox * (nx - p2)
            while px <= ex:
                p01, p10 = px + ox1, px + oy1
                p11 = p10 + ox1
                i00, i01 = enc(int(buf[px]), int(buf[p01]))
                i10, i11 = enc(int(buf[p10]), int(buf[p11]))
                buf[px], buf[p10] = enc(i00, i10)
                buf[p01], buf[p11] = enc(i01, i11)
                px += ox2
            if nx & p:
                p10 = px + oy1
                buf[px], buf[p10] = enc(int(buf[px]), int(buf[p10]))
            py += oy2
        if ny & p:
            px, ex = py, py + ox * (nx - p2)
            while px <= ex:
                p01 = px + ox1
                buf[px], buf[p01] = enc(int(buf[px]), int(buf[p01]))
                px += ox2
        p, p2 = p2, p2 << 1


# ---- Huffman
class BitWriter:
    def __init__(self):
        self.out, self.acc, self.n, self.bits = bytearray(), 0, 0, 0

    def put(self, nbits, value):
        self.acc = (self.acc << nbits) | (value & ((1 << nbits) - 1))
        self.n += nbits
        self.bits += nbits
        while self.n >= 8:
            self.n -= 8
            self.out.append((self.acc >> self.n) & 0xFF)
        self.acc &= (1 << self.n) - 1

    def flush(self):
        if self.n:
            self.out.append((self.acc << (8 - self.n)) & 0xFF)
            self.acc = self.n = 0
        return bytes(self.out)


def huf_compress(words):
    """words: sequence of uint16 -> bytes of the Huffman stage (20-byte header + packed table + data)"""
    if len(words) == 0:
        return b""
    freq = {}
    for w in words:
        freq[int(w)] = freq.get(int(w), 0) + 1
    im, iM = min(freq), max(freq) + 1            # iM: the pseudo-symbol "repeat the previous word"
    # how often a run would be emitted (decides nothing about validity; gives the symbol a sensible code)
    runs = 0
    k = 0
    while k < len(words):
        j = k
        while j + 1 < len(words) and words[j + 1] == words[k] and j - k < 255:
            j += 1
        if j - k >= 4:
            runs += 1
        k = j + 1
    freq[iM] = max(runs, 1)
    heap = [(f, s, (s,)) for s, f in sorted(freq.items())]
    heapq.heapify(heap)
    length = {s: 0 for s in freq}
    if len(heap) == 1:
        length[heap[0][1]] = 1
    while len(heap) > 1:
        f1, s1, m1 = heapq.heappop(heap)
        f2, s2, m2 = heapq.heappop(heap)
        for s in m1 + m2:
            length[s] += 1
        heapq.heappush(heap, (f1 + f2, min(s1, s2), m1 + m2))
    assert max(length.values()) <= 58
    # canonical codes: longest length starts at 0, each shorter length at (start + count) >> 1 of the next longer one
    count = [0] * 59
    for l in length.values():
        count[l] += 1
    base, c = [0] * 59, 0
    for l in range(58, 0, -1):
        nc = (c + count[l]) >> 1
        base[l] = c
        c = nc
    code = {}
    for s in sorted(length):
        l = length[s]
        code[s] = (l, base[l])
        base[l] += 1
    # the table: 6 bits per length, zero runs packed
    tw = BitWriter()
    s = im
    while s <= iM:
        l = length.get(s, 0)
        if l == 0:
            z = 1
            while s + z <= iM and length.get(s + z, 0) == 0 and z < 255 + 6:
                z += 1
            if z >= 6:
                tw.put(6, 63)
                tw.put(8, z - 6)
                s += z
                continue
            if z >= 2:
                tw.put(6, 59 + z - 2)
                s += z
                continue
        tw.put(6, l)
        s += 1
    table = tw.flush()
    dw = BitWriter()
    k = 0
    while k < len(words):
        w = int(words[k])
        j = k
        while j + 1 < len(words) and words[j + 1] == words[k] and j - k < 255:
            j += 1
        run = j - k                                  # additional repeats
        lw, cw = code[w]
        lr, cr = code[iM]
        if lw + lr + 8 < lw * run:
            dw.put(lw, cw)
            dw.put(lr, cr)
            dw.put(8, run)
        else:
            for _ in range(run + 1):
                dw.put(lw, cw)
        k = j + 1
    n_bits = dw.bits
    data = dw.flush()
    return struct.pack("<5I", im, iM, len(table), n_bits, 0) + table + data


def piz_block(rows, words_per_pixel, width):
    """rows: list over scan lines of lists over channels of uint16 arrays (width * words) -> compressed block bytes"""
    lines = len(rows)
    tmp, starts = [], []
    for c, wpp in enumerate(words_per_pixel):
        starts.append(len(tmp))
        for y in range(lines):
            tmp.extend(int(v) for v in rows[y][c])
    buf = np.array(tmp, dtype=np.int64)
    bitmap = bytearray(8192)
    for v in set(tmp):
        bitmap[v >> 3] |= 1 << (v & 7)
    bitmap[0] &= ~1 & 0xFF                           # zero is implied
    nz = [i for i in range(8192) if bitmap[i]]
    min_nz, max_nz = (nz[0], nz[-1]) if nz else (8191, 0)
    lut, k = np.zeros(65536, np.int64), 0
    for i in range(65536):
        if i == 0 or (bitmap[i >> 3] & (1 << (i & 7))):
            lut[i] = k
            k += 1
    max_value = k - 1
    buf = lut[buf]
    for c, wpp in enumerate(words_per_pixel):
        for j in range(wpp):
            wav2_encode(buf, starts[c] + j, width, wpp, lines, width * wpp, max_value)
    huf = huf_compress([int(v) for v in buf])
    out = struct.pack("<HH", min_nz, max_nz)
    if min_nz <= max_nz:
        out += bytes(bitmap[min_nz:max_nz + 1])
    return out + struct.pack("<i", len(huf)) + huf


def write_piz_exr(path, planes, names, half=False):
    """planes: dict name -> (H, W) float32; channels in alphabetical order as OpenEXR stores them"""
    names = sorted(names)
    H, W = planes[names[0]].shape
    head = struct.pack("<II", 20000630, 2)

    def attr(name, typ, payload):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<I", len(payload)) + payload
    ch = b"".join(n.encode() + b"\0" + struct.pack("<iBBBBii", 1 if half else 2, 0, 0, 0, 0, 1, 1) for n in names) + b"\0"
    head += attr("channels", "chlist", ch) + attr("compression", "compression", b"\x04")
    head += attr("dataWindow", "box2i", struct.pack("<4i", 0, 0, W - 1, H - 1)) + attr("displayWindow", "box2i", struct.pack("<4i", 0, 0, W - 1, H - 1))
    head += attr("lineOrder", "lineOrder", b"\0") + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    head += attr("screenWindowCenter", "v2f", struct.pack("<2f", 0, 0)) + attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    wpp = [1 if half else 2] * len(names)
    blocks = []
    for y0 in range(0, H, 32):
        rows = []
        for y in range(y0, min(H, y0 + 32)):
            row = []
            for n in names:
                v = planes[n][y]
                row.append(v.astype(np.float16).view(np.uint16) if half else v.astype(np.float32).view(np.uint16))  # (a float: low word first)
            rows.append(row)
        raw_size = sum(len(r) * 2 for row in rows for r in row)
        comp = piz_block(rows, wpp, W)
        if len(comp) >= raw_size:                        # the library stores a block that does not shrink as it is
            comp = b"".join(np.asarray(r, np.uint16).tobytes() for row in rows for r in row)
        blocks.append((y0, comp))
    table_at = len(head)
    off = table_at + 8 * len(blocks)
    table, body = b"", b""
    for y0, comp in blocks:
        table += struct.pack("<Q", off + len(body))
        body += struct.pack("<ii", y0, len(comp)) + comp
    with open(path, "wb") as f:
        f.write(head + table + body)


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(11)

    def image(H, W):
        y, x = np.mgrid[0:H, 0:W].astype(np.float32)
        base = 0.5 + 0.4 * np.sin(x * 0.21) * np.cos(y * 0.13)
        planes = {"R": base + 0.05 * rng.standard_normal((H, W)).astype(np.float32), "G": base * 0.5, "B": np.where(x > W // 2, 0.25, base).astype(np.float32),
                  "A": np.ones((H, W), np.float32)}
        planes["R"][H // 3:H // 3 + 4, :] = 0.0          # rows of zeros and a constant plane: the run-length symbol
        planes["G"][0, 0] = np.float32(np.inf)
        planes["G"][0, 1] = np.float32(-1e-30)
        return {k: v.astype(np.float32) for k, v in planes.items()}
    cases = {"float_35x37": (image(37, 35), False), "half_21x70": (image(70, 21), True), "float_16x16_flat": ({k: np.full((16, 16), 0.125, np.float32) for k in "ABGR"}, False)}
    for name, (planes, half) in cases.items():
        write_piz_exr(os.path.join(OUT, name + ".exr"), planes, list(planes), half)
        names = sorted(planes)
        ref = np.stack([planes[n].astype(np.float16).astype(np.float32) if half else planes[n] for n in names])
        ref.astype("<f4").tofile(os.path.join(OUT, name + ".f32"))
        print(name, os.path.getsize(os.path.join(OUT, name + ".exr")), "bytes")


if __name__ == "__main__":
    main()

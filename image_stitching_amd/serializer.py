"""The reference's stage checkpoint files (image_stitching/serializer.cpp:38-193; written / read at
image_stitching/image_stitching.cpp:651-720): ``cams.data`` (one camera per line,
``aspect@focal@ppx@ppy@<t>@<R>``) and ``indices.data`` (one index per line).  Python mirror of host/serializer.cpp
with the reference's function names; the text format is identical, so a checkpoint written by the reference feeds
the warp + blend stages of this package and vice versa.

<matrix> = "[" + every element followed by "," (or ";" at a row end) + "]"; elements are printed like C++
``operator<<`` does by default (``%g`` with 6 significant digits) and read back into float32 (CV_32F)."""
import numpy as np


def _fmt(v):
    return "%g" % float(v)      # ostream default: precision 6, neither fixed nor scientific


def serializeMatrix(m):
    m = np.asarray(m)
    if m.ndim == 1:
        m = m.reshape(-1, 1)
    out = ["["]
    for r in range(m.shape[0]):
        for c in range(m.shape[1]):
            out.append(_fmt(m[r, c]) + (";" if c == m.shape[1] - 1 else ","))
    out.append("]")
    return "".join(out)


def deserializeMatrix(s):
    if len(s) < 2 or s[0] != "[" or "]" not in s:
        raise ValueError("matrix text must look like [a,b;c,d;]")
    body = s[1:s.index("]")]
    rows = [r for r in body.split(";") if r != ""]
    vals = [[float(x) for x in r.split(",")] for r in rows]
    if len({len(r) for r in vals}) != 1:
        raise ValueError("ragged matrix text")
    return np.array(vals, dtype=np.float32)


def serializeCameraParams(cams, path="./cams.data"):
    """cams: iterable of dicts with aspect, focal, ppx, ppy, t (3), R (3x3)."""
    with open(path, "w") as f:
        for c in cams:
            t = np.asarray(c.get("t", np.zeros(3)), np.float64).reshape(3, 1)
            f.write("@".join([_fmt(c.get("aspect", 1.0)), _fmt(c["focal"]), _fmt(c["ppx"]), _fmt(c["ppy"]), serializeMatrix(t),
                              serializeMatrix(np.asarray(c["R"]))]) + "\n")


def deserializeCameraParams(path="./cams.data"):
    cams = []
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if not line:
                continue
            parts = line.split("@", 5)
            if len(parts) != 6:
                raise ValueError("cams.data: a line needs 6 '@'-separated fields")
            t, R = deserializeMatrix(parts[4]), deserializeMatrix(parts[5])
            if t.size != 3 or R.shape != (3, 3):
                raise ValueError("cams.data: t must have 3 elements and R be 3x3")
            cams.append(dict(aspect=float(parts[0]), focal=float(parts[1]), ppx=float(parts[2]), ppy=float(parts[3]),
                             t=t.reshape(3).astype(np.float64), R=R))
    return cams


def serializeIndices(indices, path="./indices.data"):
    with open(path, "w") as f:
        for i in indices:
            f.write("%d\n" % int(i))


def deserializeIndices(path="./indices.data"):
    with open(path) as f:
        return [int(line) for line in f if line.strip()]


def camera_from_checkpoint(c, width, height):
    """checkpoint entry -> the camera dict the warp / blend stages take (K from focal, aspect, ppx, ppy)."""
    K = np.array([[c["focal"], 0, c["ppx"]], [0, c["focal"] * c["aspect"], c["ppy"]], [0, 0, 1]], np.float64)
    return dict(width=width, height=height, K=K, R=np.asarray(c["R"], np.float64), focal=c["focal"])


def exif_image_description(data):
    """EXIF ImageDescription (tag 0x010E, ASCII) of a JPEG file's bytes, or None -- the tag walk of the reference's camera loader
    (image_stitching.cpp:344-347, :411-417) without libexif; host/exif.cpp is the C++ twin.  IFD0, IFD1, then the Exif sub-IFD; the
    last occurrence wins; cut to 1022 characters as exif_entry_get_value(ee, buf, 1023) does."""
    d = bytes(data)
    n = len(d)
    if n < 4 or d[0] != 0xFF or d[1] != 0xD8:
        return None
    o = 2
    while o + 4 <= n:
        if d[o] != 0xFF:
            return None
        m = d[o + 1]
        if m == 0xFF:
            o += 1
            continue
        if m in (0xD8, 0x01) or 0xD0 <= m <= 0xD7:
            o += 2
            continue
        if m in (0xD9, 0xDA):
            return None
        ln = (d[o + 2] << 8) | d[o + 3]
        if ln < 2 or o + 2 + ln > n:
            return None
        if m == 0xE1 and ln >= 16 and d[o + 4:o + 10] == b"Exif\0\0":
            t = d[o + 10:o + 2 + ln]
            if len(t) < 8 or t[:2] not in (b"II", b"MM"):
                return None
            bo = "little" if t[:2] == b"II" else "big"
            u16 = lambda p: int.from_bytes(t[p:p + 2], bo)
            u32 = lambda p: int.from_bytes(t[p:p + 4], bo)
            if u16(2) != 42:
                return None
            found = [None]

            def walk(off, want_exif):
                if not off or off + 2 > len(t):
                    return 0, 0
                cnt = u16(off)
                if off + 2 + cnt * 12 + 4 > len(t):
                    return 0, 0
                exif = 0
                for i in range(cnt):
                    e = off + 2 + i * 12
                    tag, typ, count = u16(e), u16(e + 2), u32(e + 4)
                    if tag == 0x8769 and want_exif:
                        exif = u32(e + 8)
                    if tag != 0x010E or typ != 2:
                        continue
                    vo = e + 8 if count <= 4 else u32(e + 8)
                    if vo + count > len(t):
                        continue
                    v = t[vo:vo + count].split(b"\0")[0][:1022]
                    found[0] = v.decode("latin-1")
                return exif, u32(off + 2 + cnt * 12)
            exif, nxt = walk(u32(4), True)
            walk(nxt, False)
            walk(exif, False)
            return found[0]
        o += 2 + ln
    return None

"""Image output of the reference's front end (src/main.rs:28-53) as written by the host library: the EXR and the PNG are
read back with independent minimal readers and compared value for value."""
import struct
import zlib

import numpy as np

import pbrs_amd

f32 = np.float32


def read_exr(path):
    """Single-part, uncompressed, scanline OpenEXR with FLOAT channels (what pbrs_host_write_exr emits)."""
    d = open(path, "rb").read()
    assert struct.unpack_from("<II", d, 0) == (20000630, 2)
    p, attrs = 8, {}
    while d[p] != 0:
        e = d.index(b"\0", p)
        name = d[p:e].decode()
        p = e + 1
        e = d.index(b"\0", p)
        typ = d[p:e].decode()
        p = e + 1
        (size,) = struct.unpack_from("<I", d, p)
        attrs[name] = (typ, d[p + 4:p + 4 + size])
        p += 4 + size
    p += 1
    assert attrs["compression"][1] == b"\0" and attrs["lineOrder"][1] == b"\0"
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    chans, q, cl = [], 0, attrs["channels"][1]
    while cl[q] != 0:
        e = cl.index(b"\0", q)
        chans.append(cl[q:e].decode())
        assert struct.unpack_from("<I", cl, e + 1)[0] == 2  # FLOAT
        q = e + 1 + 16
    assert chans == ["B", "G", "R"]
    offsets = struct.unpack_from("<%dQ" % h, d, p)
    img = np.empty((h, w, 3), dtype=f32)
    for y in range(h):
        yy, nbytes = struct.unpack_from("<iI", d, offsets[y])
        assert yy == y and nbytes == 12 * w
        row = np.frombuffer(d, dtype="<f4", count=3 * w, offset=offsets[y] + 8).reshape(3, w)
        img[y, :, 2], img[y, :, 1], img[y, :, 0] = row[0], row[1], row[2]
    return img


def read_png(path):
    d = open(path, "rb").read()
    assert d[:8] == b"\x89PNG\r\n\x1a\n"
    p, idat, w, h = 8, b"", 0, 0
    while p < len(d):
        (n,) = struct.unpack_from(">I", d, p)
        t = d[p + 4:p + 8]
        body = d[p + 8:p + 8 + n]
        assert struct.unpack_from(">I", d, p + 8 + n)[0] == zlib.crc32(t + body) & 0xFFFFFFFF
        if t == b"IHDR":
            w, h, depth, ctype = struct.unpack_from(">IIBB", body)
            assert (depth, ctype) == (8, 2)
        elif t == b"IDAT":
            idat += body
        p += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + 3 * w)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 3)


def test_exr_keeps_every_float_bit(tmp_path):
    rs = np.random.RandomState(0)
    img = (rs.standard_normal((13, 17, 3)) * np.exp(rs.uniform(-20, 20, (13, 17, 1)))).astype(f32)
    img[0, 0] = [np.inf, -0.0, np.nan]
    pbrs_amd.write_image(tmp_path / "a.exr", img)
    back = read_exr(tmp_path / "a.exr")
    assert (back.view(np.uint32) == img.view(np.uint32)).all()


def test_png_is_sqrt_gamma_then_saturating_cast(tmp_path):
    rs = np.random.RandomState(1)
    img = rs.uniform(-0.2, 1.5, (9, 11, 3)).astype(f32)
    img[0, 0] = [np.nan, 0.0, 1.0]
    img[0, 1] = [4.0, 0.25, 1e-9]
    pbrs_amd.write_image(tmp_path / "a.png", img)
    got = read_png(tmp_path / "a.png")
    with np.errstate(invalid="ignore"):
        g = np.sqrt(img)  # Color::gamma_encode; sqrt of a negative is NaN -> 0 like any NaN (color.rs:13-23)
    want = np.where(g > 1.0, 255, np.where(g >= 0.0, np.nan_to_num(g * f32(255.0), nan=0.0), 0)).astype(np.uint8)
    assert (got == want).all()
    assert got[0, 0].tolist() == [0, 0, 255] and got[0, 1].tolist() == [255, 127, 0]

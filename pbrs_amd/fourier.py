"""`.bsdf` tables of the Fourier BSDF (geometry/src/fourier.rs:13-221): reading and writing the SCATFUN file format, and
building small synthetic tables by projecting a given isotropic BSDF onto the format's spline x Fourier basis.

The format (layerlab, Jakob et al. 2014) stores, for every pair of zenith cosines (mu_i, mu_o) of a discretisation `mu`,
the cosine series a_k of  f(mu_i, mu_o, phi) * |mu_i|  in the azimuth difference phi (only as many coefficients as the
pair needs, `m`), for one channel (luminance) or three (luminance, red, blue); `cdf[o][i]` is the running integral of the
order-0 coefficients over mu_i for a fixed mu_o, which `sample_catmull_rom_2d` inverts.  The reference's sign conventions
(FourierBSDF::eval, :300-360): mu_i = -cos(theta_i), mu_o = cos(theta_o), phi = azimuth of wo against -wi.
"""
import struct

import numpy as np

f32 = np.float32
HEADER = struct.Struct("<7sBIiiiiiiiif2f2f")  # geometry/src/fourier.rs:55-72 (64 bytes)
assert HEADER.size == 64


class FourierTable:
    def __init__(self, mu, cdf, offset_and_length, a, n_channels, eta=1.0):
        self.mu = np.asarray(mu, dtype=f32)
        self.cdf = np.asarray(cdf, dtype=f32).reshape(len(self.mu), len(self.mu))
        self.offset_and_length = np.asarray(offset_and_length, dtype=np.int32).reshape(-1, 2)
        self.a = np.asarray(a, dtype=f32)
        self.n_channels = int(n_channels)
        self.eta = float(eta)

    @property
    def m_max(self):
        return int(self.offset_and_length[:, 1].max())


def parse_header(raw):
    """read_header (:74-96) on the first 64 bytes of a file."""
    ident, version, flags, n_mu, n_coeffs, m_max, n_channels, n_bases, n_meta, n_par, n_parv, eta, *alpha = HEADER.unpack_from(raw, 0)
    if ident != b"SCATFUN" or version != 1 or flags != 1:
        raise ValueError("not a version-1 SCATFUN BSDF file")  # read_header's asserts (:84-88)
    return {"n_mu": n_mu, "n_coeffs": n_coeffs, "m_max": m_max, "n_channels": n_channels, "n_bases": n_bases, "eta": eta}


def read_bsdf(path):
    """FourierTable::from_file (:167-221): header, mu, cdf, (offset, length) pairs, coefficients."""
    with open(path, "rb") as f:
        raw = f.read()
    h = parse_header(raw)
    n_mu, n_coeffs, n_channels, eta = h["n_mu"], h["n_coeffs"], h["n_channels"], h["eta"]
    at = 64
    mu = np.frombuffer(raw, dtype="<f4", count=n_mu, offset=at)
    at += 4 * n_mu
    cdf = np.frombuffer(raw, dtype="<f4", count=n_mu * n_mu, offset=at)
    at += 4 * n_mu * n_mu
    ol = np.frombuffer(raw, dtype="<i4", count=2 * n_mu * n_mu, offset=at)
    at += 8 * n_mu * n_mu
    a = np.frombuffer(raw, dtype="<f4", count=n_coeffs, offset=at)
    return FourierTable(mu, cdf, ol, a, n_channels, eta)


def write_bsdf(path, t):
    n_mu = len(t.mu)
    head = HEADER.pack(b"SCATFUN", 1, 1, n_mu, len(t.a), t.m_max, t.n_channels, 1, 0, 0, 0, t.eta, 0.0, 0.0, 0.0, 0.0)
    with open(path, "wb") as f:
        f.write(head)
        f.write(np.asarray(t.mu, dtype="<f4").tobytes())
        f.write(np.asarray(t.cdf, dtype="<f4").tobytes())
        f.write(np.asarray(t.offset_and_length, dtype="<i4").tobytes())
        f.write(np.asarray(t.a, dtype="<f4").tobytes())


def project(bsdf, n_mu=24, m=12, n_channels=3, n_phi=256, tol=1e-4):
    """A table for the isotropic BSDF `bsdf(mu_i, mu_o, cos_phi) -> (y, r, b)` (arrays broadcast; the reference's conventions
    for the three arguments): cosine-series coefficients of bsdf * |mu_i| by the trapezoid rule on n_phi azimuths, truncated
    per pair where they fall below tol of the order-0 term (so that pairs carry different lengths, as real files do);
    mu = the Gauss-Lobatto-like nodes real files use, -1 .. 1 with both hemispheres; cdf by the trapezoid rule over mu_i."""
    neg = -np.cos(np.linspace(0.0, np.pi / 2, n_mu // 2, endpoint=True))  # -1 .. 0, ascending
    neg[-1] = -1e-4  # catmull_rom_weights needs distinct knots around the horizon
    mu = np.concatenate([neg, -neg[::-1]]).astype(f32)
    n = len(mu)
    phi = (np.arange(n_phi) + 0.5) * (np.pi / n_phi)
    ks = np.arange(m)
    basis = np.cos(np.outer(ks, phi))  # [k][phi]
    norm = np.where(ks == 0, 1.0 / n_phi, 2.0 / n_phi)
    coeffs, ol = [], []
    a0 = np.zeros((n, n))
    offset = 0
    for o in range(n):
        for i in range(n):
            vals = np.stack(np.broadcast_arrays(*bsdf(np.float64(mu[i]), np.float64(mu[o]), np.cos(phi))), 0)[:n_channels]
            ak = (vals * abs(float(mu[i]))) @ basis.T * norm  # [channel][k]
            if ak[0, 0] <= 0.0:
                ol.append((offset, 0))
                continue
            keep = np.nonzero(np.abs(ak).max(axis=0) > tol * ak[0, 0])[0]
            length = int(keep.max()) + 1
            a0[o, i] = ak[0, 0]
            coeffs.append(ak[:, :length].astype(f32).reshape(-1))
            ol.append((offset, length))
            offset += n_channels * length
    cdf = np.zeros((n, n))
    for o in range(n):
        for i in range(1, n):
            cdf[o, i] = cdf[o, i - 1] + 0.5 * (a0[o, i] + a0[o, i - 1]) * float(mu[i] - mu[i - 1])
    a = np.concatenate(coeffs) if coeffs else np.zeros(0, dtype=f32)
    return FourierTable(mu, cdf, np.array(ol, dtype=np.int32), a, n_channels)


def lambert_like(albedo=(0.6, 0.5, 0.4)):
    """A diffuse reflector with a slight lobe towards the mirror direction (so that higher orders are present): reflection
    only (mu_i * mu_o < 0 in the reference's convention), the same on both faces."""
    r, g, b = albedo
    y_w = 0.212671 * r + 0.715160 * g + 0.072169 * b

    def f(mu_i, mu_o, cos_phi):
        refl = 1.0 if mu_i * mu_o < 0.0 else 0.0  # either face
        lobe = 1.0 + 0.8 * np.maximum(cos_phi, 0.0) ** 4 * (1.0 - abs(mu_o)) * (1.0 - abs(mu_i))
        base = refl * lobe / np.pi
        return base * y_w, base * r, base * b

    return f


def translucent(albedo=(0.5, 0.5, 0.5), through=0.5):
    """lambert_like plus a diffuse transmitted part: FourierBSDF::sample reaches its `todo!()` (:423-428) whenever it draws
    a direction on the far side."""
    base = lambert_like(albedo)

    def f(mu_i, mu_o, cos_phi):
        y, r, b = base(mu_i, mu_o, cos_phi)
        t = through / np.pi if mu_i * mu_o > 0.0 else 0.0
        return y + t + 0.0 * cos_phi, r + t + 0.0 * cos_phi, b + t + 0.0 * cos_phi

    return f

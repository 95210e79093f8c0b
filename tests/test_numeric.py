"""include/pbrs_numeric.h (the f32 libm contract shared by host, kernels and oracle) against the platform
libm, and the RNG contract of SURVEY.md Appendix B."""
import numpy as np
import pytest

from oracle.binding import numeric_eval, rng_stream


def ulp_err(got, want64):
    want32 = want64.astype(np.float32)
    spacing = np.spacing(np.abs(want32)).astype(np.float64)
    spacing[spacing == 0] = np.finfo(np.float32).tiny
    return np.abs(got.astype(np.float64) - want64) / spacing


RS = np.random.RandomState(1)
CASES = [
    ("sin", RS.uniform(-50, 50, 200000), np.sin, 2.0),
    ("cos", RS.uniform(-50, 50, 200000), np.cos, 2.0),
    ("atan", RS.standard_normal(200000) * 20, np.arctan, 3.0),
    ("asin", RS.uniform(-1, 1, 200000), np.arcsin, 3.0),
    ("acos", RS.uniform(-1, 1, 200000), np.arccos, 3.0),
    ("exp", RS.uniform(-80, 80, 200000), np.exp, 2.0),
    ("ln", np.exp(RS.uniform(-80, 80, 200000)), np.log, 2.0),
    ("sqrt", np.exp(RS.uniform(-80, 80, 200000)), np.sqrt, 0.5),
]


@pytest.mark.parametrize("fn,x,ref,max_ulp", CASES, ids=[c[0] for c in CASES])
def test_matches_libm_within_ulps(fn, x, ref, max_ulp):
    x = x.astype(np.float32)
    got = numeric_eval(fn, x)
    err = ulp_err(got, ref(x.astype(np.float64)))
    if fn in ("sin", "cos"):  # absolute accuracy near the zeros of sin/cos is bounded by the 3-part pi/4 reduction
        small = np.abs(ref(x.astype(np.float64))) < 1e-3
        assert np.abs(got.astype(np.float64) - ref(x.astype(np.float64)))[small].max() < 1e-7
        err = err[~small]
    assert err.max() <= max_ulp, (fn, float(err.max()))


def test_tan_atan2_hypot():
    x = RS.uniform(-1.5, 1.5, 100000).astype(np.float32)
    assert ulp_err(numeric_eval("tan", x), np.tan(x.astype(np.float64))).max() <= 4.0
    a = (RS.standard_normal(100000) * 5).astype(np.float32)
    b = (RS.standard_normal(100000) * 5).astype(np.float32)
    got = numeric_eval("atan2", a, b)
    assert np.abs(got.astype(np.float64) - np.arctan2(a.astype(np.float64), b.astype(np.float64))).max() < 1e-6
    assert ulp_err(numeric_eval("hypot", a, b), np.hypot(a.astype(np.float64), b.astype(np.float64))).max() <= 1.5


def test_special_values():
    inf = np.float32(np.inf)
    assert numeric_eval("exp", [-inf, inf, 0.0]).tolist() == [0.0, np.inf, 1.0]
    assert numeric_eval("ln", [0.0, 1.0, inf]).tolist() == [-np.inf, 0.0, np.inf]
    assert np.isnan(numeric_eval("ln", [-1.0])[0])
    assert np.isnan(numeric_eval("acos", [1.5])[0])
    assert numeric_eval("acos", [1.0, -1.0]).tolist() == [0.0, np.float32(np.pi)]
    assert numeric_eval("atan2", [0.0, 1.0, -1.0], [0.0, 0.0, 0.0]).tolist() == [0.0, np.float32(np.pi / 2), -np.float32(np.pi / 2)]
    assert numeric_eval("fract", [1.75, -1.75, 3.0]).tolist() == [0.75, -0.75, 0.0]
    assert numeric_eval("floor", [1.75, -1.75, -3.0]).tolist() == [1.0, -2.0, -3.0]
    # compiler-rt __powisf2 association: powi(x, 5) = x * (x^2)^2
    x = np.float32(1.1)
    assert numeric_eval("powi", [x], [5.0])[0] == np.float32(x * np.float32(np.float32(x * x) * np.float32(x * x)))


def test_division_and_sqrt_are_ieee():
    a = (RS.standard_normal(100000) * np.exp(RS.uniform(-30, 30, 100000))).astype(np.float32)
    b = (RS.standard_normal(100000) * np.exp(RS.uniform(-30, 30, 100000))).astype(np.float32)
    assert (numeric_eval("div", a, b).view(np.uint32) == (a / b).view(np.uint32)).all()
    assert (numeric_eval("sqrt", np.abs(a)).view(np.uint32) == np.sqrt(np.abs(a)).view(np.uint32)).all()


def test_rng_contract():
    """PCG32 keyed by (seed, pixel, sample); f32 = (u32 >> 8) * 2^-24 in [0, 1) (rand 0.8 `Standard`)."""
    a = rng_stream(1, 7, 3, 4096)
    assert (a >= 0).all() and (a < 1).all()
    assert ((a * 2 ** 24) == np.round(a * 2 ** 24)).all()  # 24-bit mantissa grid
    assert (rng_stream(1, 7, 3, 16) == a[:16]).all()
    assert (rng_stream(1, 7, 4, 16) != a[:16]).any() and (rng_stream(1, 8, 3, 16) != a[:16]).any() and (rng_stream(2, 7, 3, 16) != a[:16]).any()
    # golden: first draws of stream (seed=1, pixel=0, sample=0); pins the contract across rounds
    golden = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "rng_seed1_px0_s0.npy"))
    assert (rng_stream(1, 0, 0, len(golden)).view(np.uint32) == golden.view(np.uint32)).all()
    # mean / uniformity sanity
    big = rng_stream(123, 5, 9, 1 << 16)
    assert abs(big.mean() - 0.5) < 5e-3

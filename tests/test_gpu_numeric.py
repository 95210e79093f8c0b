"""gfx950 evaluates include/pbrs_numeric.h (and IEEE divide / sqrt) bit for bit like x86: the premise of
"radiance matches the CPU reference at matched seeds".  Calls go through the C ABI (pbrs_numeric_eval)."""
import numpy as np
import pytest

from oracle.binding import numeric_eval

pytestmark = pytest.mark.gpu
RS = np.random.RandomState(3)
N = 1 << 18

UNARY = {
    "sin": lambda: RS.uniform(-60, 60, N), "cos": lambda: RS.uniform(-60, 60, N), "tan": lambda: RS.uniform(-8, 8, N),
    "atan": lambda: RS.standard_normal(N) * 30, "asin": lambda: RS.uniform(-1, 1, N), "acos": lambda: RS.uniform(-1.001, 1.001, N),
    "exp": lambda: RS.uniform(-110, 95, N), "ln": lambda: np.exp(RS.uniform(-100, 88, N)), "sqrt": lambda: np.exp(RS.uniform(-100, 88, N)),
    "fract": lambda: RS.standard_normal(N) * 1e3, "floor": lambda: RS.standard_normal(N) * 1e3,
}


@pytest.mark.parametrize("fn", sorted(UNARY))
def test_unary_bit_exact(gpu_ctx, fn):
    x = UNARY[fn]().astype(np.float32)
    x[:8] = [0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, 1e-45]
    cpu, gpu = numeric_eval(fn, x), gpu_ctx.numeric_eval(fn, x)
    nan = np.isnan(cpu)
    assert (nan == np.isnan(gpu)).all()
    assert (cpu.view(np.uint32)[~nan] == gpu.view(np.uint32)[~nan]).all()


@pytest.mark.parametrize("fn", ["div", "hypot", "atan2"])
def test_binary_bit_exact(gpu_ctx, fn):
    x = (RS.standard_normal(N) * np.exp(RS.uniform(-40, 40, N))).astype(np.float32)
    y = (RS.standard_normal(N) * np.exp(RS.uniform(-40, 40, N))).astype(np.float32)
    x[:6] = [0.0, 1.0, -1.0, 0.0, np.inf, 1e-40]
    y[:6] = [0.0, 0.0, 0.0, -1.0, 2.0, 3.0]
    cpu, gpu = numeric_eval(fn, x, y), gpu_ctx.numeric_eval(fn, x, y)
    nan = np.isnan(cpu)
    assert (nan == np.isnan(gpu)).all()
    assert (cpu.view(np.uint32)[~nan] == gpu.view(np.uint32)[~nan]).all()


def test_denormal_division_is_not_flushed(gpu_ctx):
    x = np.array([1e-38, 3e-39, 1.4e-45], dtype=np.float32)
    y = np.array([4.0, 2.0, 1.0], dtype=np.float32)
    assert (gpu_ctx.numeric_eval("div", x, y).view(np.uint32) == (x / y).view(np.uint32)).all()


def test_box_test_quotient_is_the_correctly_rounded_division(gpu_ctx):
    """device/traverse.h: q0 = nn nr; e = fma(d, q0, nn); q = fma(e, nr, q0) on nn = -n, nr = -RN(1 / d) must be n / d bit for bit
    wherever the division-free box test is used (|d| in [2^-40, 2^40], n zero or of moderate size).  The proof is the exhaustive
    run of tools/microbench/div_exhaustive.hip over all 2^46 significand pairs (profiles/r03_div_exhaustive.log); this test keeps
    the kernel honest about it on random operands and on operands built to sit next to f32 rounding boundaries (n / d within
    2^-47 of a midpoint between two neighbouring floats, the closest a ratio of two 24-bit significands can get)."""
    n = (RS.standard_normal(N) * np.exp(RS.uniform(-30, 30, N))).astype(np.float32)
    d = (np.where(RS.rand(N) < 0.5, -1.0, 1.0) * np.exp(RS.uniform(np.log(2.0 ** -40), np.log(2.0 ** 40), N))).astype(np.float32)
    n[:4] = [0.0, -0.0, 1.0, 3.0]
    d[:4] = [3.0, -7.0, 3.0, 1.0]
    # adversarial: for an odd 24-bit d take the odd 25-bit m (a midpoint m 2^-24 between two floats of [1, 2)) with
    # m d = +-1 (mod 2^24) and n = (m d -+ 1) 2^-24: then |n / d - m 2^-24| = 2^-24 / d, about 2^-47 relative, the closest a
    # ratio of two 24-bit significands gets to a rounding boundary
    n2, d2 = [], []
    for dv in (RS.randint(1 << 23, 1 << 24, 1 << 13).astype(np.int64) | 1).tolist():
        inv = pow(dv, -1, 1 << 24)
        for sign in (1, -1):
            m = (sign * inv) % (1 << 24) + (1 << 24)
            nv = (m * dv - sign) >> 24
            if (1 << 23) <= nv < (1 << 24):
                assert (m * dv - sign) % (1 << 24) == 0
                n2.append(nv)
                d2.append(dv)
    assert len(n2) > 1000
    scale_n = np.float32(2.0) ** RS.randint(-20, 20, len(n2)).astype(np.float32)
    scale_d = np.float32(2.0) ** RS.randint(-60, 16, len(n2)).astype(np.float32)  # |d| stays inside [2^-40, 2^40]
    n2 = np.array(n2, dtype=np.float32) * scale_n * np.where(RS.rand(len(n2)) < 0.5, -1, 1).astype(np.float32)
    d2 = np.array(d2, dtype=np.float32) * scale_d
    nn, dd = np.concatenate([n, n2]), np.concatenate([d, d2])
    want = nn / dd
    got = gpu_ctx.numeric_eval("box_quotient", nn, dd)
    finite = np.isfinite(want) & ((want == 0) | (np.abs(want) > 1e-30))  # no f32 underflow: the box test's guard excludes it
    assert (want.view(np.uint32)[finite] == got.view(np.uint32)[finite]).all()
    assert finite.mean() > 0.95

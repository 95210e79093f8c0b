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

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden  # noqa: E402

GOLDEN_NAMES = list(make_golden.CASES)
SEED = make_golden.SEED


def load_golden(name):
    return np.load(os.path.join(HERE, "golden", name + ".npz"))


def golden_case(name):
    return make_golden.build(name)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)

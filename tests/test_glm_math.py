"""Accuracy of the libm restatement in csrc/hrt_glm.h (the functions the reference takes from the platform
libm: sin, cos, acos, atan2, log) against numpy float64.  Bound: 4 ulp of the fp32 result (or 4 ulp of 1.0
near zeros of sin/cos, where relative error is meaningless).  The CPU == GPU bit equality of the same
kernels is checked in tests/test_gpu_parity.py::test_math_kernels_bit_exact."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def orc(built):
    from oracle import oracle_py
    return oracle_py


def _ulp_err(got, ref64, floor=0.0):
    ref32 = ref64.astype(np.float32)
    ulp = np.maximum(np.abs(np.spacing(ref32)).astype(np.float64), floor)
    return np.abs(got.astype(np.float64) - ref64) / ulp


def test_sin_cos(orc):
    r = np.random.default_rng(0)
    x = np.concatenate([r.uniform(-7, 7, 300000), r.uniform(-60, 60, 300000), np.linspace(0, 2 * np.pi, 10001)]).astype(np.float32)
    eps1 = float(np.spacing(np.float32(1.0)))
    for op, f in ((0, np.sin), (1, np.cos)):
        err = _ulp_err(orc.math_probe(op, x), f(x.astype(np.float64)), floor=eps1 / 4)
        assert err.max() < 4.0, (op, err.max(), x[err.argmax()])


def test_acos(orc):
    r = np.random.default_rng(1)
    x = np.concatenate([r.uniform(-1, 1, 500000), [1.0, -1.0, 0.0, 0.5, -0.5, 0.999999, -0.999999]]).astype(np.float32)
    err = _ulp_err(orc.math_probe(2, x), np.arccos(x.astype(np.float64)), floor=float(np.spacing(np.float32(1.0))) / 4)
    assert err.max() < 4.0, (err.max(), x[err.argmax()])
    assert np.isnan(orc.math_probe(2, np.array([1.5, -2.0], np.float32))).all()


def test_atan2(orc):
    r = np.random.default_rng(2)
    x = r.uniform(-3, 3, 500000).astype(np.float32)
    y = r.uniform(-3, 3, 500000).astype(np.float32)
    got = orc.math_probe(3, x, y)  # atan2(y, x)
    err = _ulp_err(got, np.arctan2(y.astype(np.float64), x.astype(np.float64)), floor=float(np.spacing(np.float32(1.0))) / 4)
    assert err.max() < 4.0, err.max()
    # axes
    ax = orc.math_probe(3, np.array([1, -1, 0, 0], np.float32), np.array([0, 0, 1, -1], np.float32))
    np.testing.assert_allclose(ax, [0.0, np.pi, np.pi / 2, -np.pi / 2], atol=1e-6)


def test_log(orc):
    r = np.random.default_rng(3)
    x = np.concatenate([r.uniform(0, 1, 300000), np.exp(r.uniform(-80, 80, 300000)), [1.0, 0.5, 2.0]]).astype(np.float32)
    x = x[x > 0]
    err = _ulp_err(orc.math_probe(4, x), np.log(x.astype(np.float64)), floor=float(np.spacing(np.float32(1.0))) / 4)
    assert err.max() < 4.0, (err.max(), x[err.argmax()])
    assert orc.math_probe(4, np.array([0.0], np.float32))[0] == -np.inf

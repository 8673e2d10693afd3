"""Accuracy of csrc/pt_math.h (the shared host / device math the product and the oracle's math mode 0 both use) against
mpmath at 60 digits: the claim in pt_math.h's header.  The functions are exercised through the oracle's host build; the
device build is bit-identical to it (tests/test_gpu_parity.py::test_math_device_equals_host_bitwise)."""
import math

import numpy as np
import pytest

mp = pytest.importorskip("mpmath")
mp.mp.dps = 60

FN = {"hypot": 0, "sin": 1, "cos": 2, "acos": 3, "atan2": 4, "pow5": 5}


def ulp_errors(got, exact):
    """|got - exact| in units of the last place of the exact value's binade"""
    out = []
    for g, e in zip(got, exact):
        if e == 0:
            out.append(0.0 if g == 0 else float("inf"))
            continue
        ulp = mp.mpf(2) ** (mp.floor(mp.log(abs(e), 2)) - 52)
        out.append(float(abs(mp.mpf(float(g)) - e) / ulp))
    return np.array(out)


@pytest.mark.parametrize("fn,bound", [("hypot", 1.0), ("sin", 1.0), ("cos", 1.0), ("acos", 1.0), ("atan2", 1.5), ("pow5", 0.5000001)])
def test_pt_math_accuracy_vs_mpmath(oracle, fn, bound):
    rng = np.random.default_rng(100 + FN[fn])
    n = 6000
    b = None
    if fn == "hypot":  # V3.normalize / Quaternion.normalize: components of unit-ish vectors and scene-scale offsets
        a = rng.uniform(-4, 4, n) * 10.0 ** rng.integers(-3, 4, n)
        b = rng.uniform(-4, 4, n) * 10.0 ** rng.integers(-3, 4, n)
        exact = [mp.sqrt(mp.mpf(float(x)) ** 2 + mp.mpf(float(y)) ** 2) for x, y in zip(a, b)]
    elif fn in ("sin", "cos"):  # theta = v * 2 * pi, v in [0, 1)  (shader_space.ml:56-64)
        a = np.concatenate([rng.uniform(0, 2 * math.pi, n - 200), rng.uniform(-50, 50, 200)])
        f = mp.sin if fn == "sin" else mp.cos
        exact = [f(mp.mpf(float(x))) for x in a]
    elif fn == "acos":  # acos (-n.y), |n| = 1  (sphere.ml:25-33)
        a = np.concatenate([rng.uniform(-1, 1, n - 4), [1.0, -1.0, 0.0, 0.999999999]])
        exact = [mp.acos(mp.mpf(float(x))) for x in a]
    elif fn == "atan2":  # atan2 (-n.z) n.x
        a = rng.uniform(-1, 1, n)
        b = rng.uniform(-1, 1, n)
        exact = [mp.atan2(mp.mpf(float(y)), mp.mpf(float(x))) for y, x in zip(a, b)]
    else:  # (1 - cos) ** 5.0, 1 - cos in [0, 1]  (material.ml:16-20,37)
        a = np.concatenate([rng.uniform(0, 1, n - 500), rng.uniform(-2, 2, 500)])
        exact = [mp.mpf(float(x)) ** 5 for x in a]
    got = oracle.math_vec(FN[fn], a, b)
    err = ulp_errors(got, exact)
    assert np.isfinite(err).all()
    assert err.max() <= bound, f"{fn}: max error {err.max():.3f} ulp (bound {bound}) at input {a[int(err.argmax())]!r}"


@pytest.mark.parametrize("fused,nested", [(19, 9), (20, 10)])
def test_fused_normalisation_scalars_equal_the_nested_expression(oracle, fused, nested):
    """pt_rnorm3 / pt_rnorm_frame (one range test, bare sqrt / reciprocal sequences on the device) against the expression they
    stand for, `1 / hypot x (hypot y z)` evaluated call by call through pt_hypot (affine.ml:65-68, quaternion.ml:11-15) -- bit
    for bit, on the host build: unit-scale and scene-scale components, exact zeros, scales from 1e-290 to 1e300, specials.
    (Subnormal operands are outside the claim: the nested expression rounds its inner hypot to the subnormal grid before the
    outer one sees it, the fused form scales once and keeps those bits -- both are valid evaluations; no scene has them.)"""
    rng = np.random.default_rng(fused)
    n = 400_000
    a = rng.uniform(-4, 4, n) * 10.0 ** rng.integers(-3, 4, n)
    b = rng.uniform(-4, 4, n) * 10.0 ** rng.integers(-3, 4, n)
    k = n // 8
    a[:k] = 0.0
    b[k:2 * k] = 0.0
    b[2 * k:3 * k] = a[2 * k:3 * k]
    a[3 * k:4 * k] *= 10.0 ** rng.integers(-290, 300, k)
    b[3 * k:4 * k] *= 10.0 ** rng.integers(-290, 300, k)
    s = 10.0 ** rng.integers(-290, 300, k)
    a[4 * k:5 * k] *= s
    b[4 * k:5 * k] *= s
    a[5 * k:5 * k + 5] = [np.inf, np.nan, 0.0, -0.0, -1.0]
    b[5 * k:5 * k + 5] = [1.0, 1.0, 0.0, 0.0, 0.0]
    got = oracle.math_vec(fused, a, b)
    want = oracle.math_vec(nested, a, b)
    same = (got.view(np.uint64) == want.view(np.uint64)) | (np.isnan(got) & np.isnan(want))
    bad = np.flatnonzero(~same)
    assert bad.size == 0, f"{bad.size} differ, first: a={a[bad[0]]!r} b={b[bad[0]]!r} fused={got[bad[0]]!r} nested={want[bad[0]]!r}"

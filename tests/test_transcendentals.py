"""The deterministic acos/atan2/sin/cos replacements stay within 1 fp32 ulp of libm."""
import ctypes as C
import numpy as np


def ulp_err(got, want):
    want32 = want.astype(np.float32)
    spacing = np.spacing(np.abs(want32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - want) / np.maximum(spacing, 1e-45)


def test_det_functions_close_to_libm(oracle_mod):
    L = oracle_mod.library().lib
    for n in ("ora_det_atan2f", "ora_det_acosf", "ora_det_sinf", "ora_det_cosf"):
        getattr(L, n).restype = C.c_float
    L.ora_det_atan2f.argtypes = [C.c_float, C.c_float]
    for n in ("ora_det_acosf", "ora_det_sinf", "ora_det_cosf"):
        getattr(L, n).argtypes = [C.c_float]
    rng = np.random.default_rng(0)
    y = rng.uniform(-4, 4, 4000).astype(np.float32); x = rng.uniform(-4, 4, 4000).astype(np.float32)
    got = np.array([L.ora_det_atan2f(float(a), float(b)) for a, b in zip(y, x)], np.float32)
    assert ulp_err(got, np.arctan2(y.astype(np.float64), x.astype(np.float64))).max() <= 1.0
    assert L.ora_det_atan2f(0.0, 0.0) == 0.0 and abs(L.ora_det_atan2f(0.0, -1.0) - np.float32(np.pi)) < 1e-6
    t = rng.uniform(-1, 1, 4000).astype(np.float32)
    got = np.array([L.ora_det_acosf(float(a)) for a in t], np.float32)
    # acos goes through atan2(sqrt((1-x)(1+x)), x): 1 ulp away from +-1, a few ulp next to them
    assert ulp_err(got, np.arccos(t.astype(np.float64))).max() <= 4.0
    assert L.ora_det_acosf(1.0) == 0.0 and L.ora_det_acosf(2.0) == 0.0
    a = rng.uniform(-7, 7, 4000).astype(np.float32)
    s = np.array([L.ora_det_sinf(float(v)) for v in a], np.float32); c = np.array([L.ora_det_cosf(float(v)) for v in a], np.float32)
    assert np.abs(s.astype(np.float64) - np.sin(a.astype(np.float64))).max() < 1.2e-7
    assert np.abs(c.astype(np.float64) - np.cos(a.astype(np.float64))).max() < 1.2e-7

#!/usr/bin/env python3
"""Coefficients of the GELU used by k_post_block (csrc/encoder_post.hip):

    gelu(y) = y Phi(y) = max(y, 0) - |y| * 2 ** P(|y|),   P(t) ~ log2 Phi(-t)

P is a degree-5 polynomial fitted on [0, 6] by iteratively re-weighted least squares (a minimax
fit of the error in gelu, i.e. weighted by t Phi(-t) ln 2).  Prints the coefficients (lowest
degree first) and the maximum absolute error of the fp32 evaluation over |y| <= 40 against the
float64 erf form, next to the error of the degree-10 erf polynomial the other GEMM paths use."""
import numpy as np
from scipy.special import log_ndtr, ndtr

LN2 = np.log(2.0)


def fit(deg=5, T=6.0, iters=60):
    t = np.cos(np.pi * (np.arange(6000) + 0.5) / 6000) * T / 2 + T / 2
    f = log_ndtr(-t) / LN2
    sens = np.maximum(t * np.exp(log_ndtr(-t)) * LN2, 1e-6)
    w = np.ones_like(t)
    for _ in range(iters):
        A = np.vander(t, deg + 1, increasing=True)
        W = w * sens
        coef = np.linalg.lstsq(A * W[:, None], f * W, rcond=None)[0]
        err = np.abs((A @ coef - f) * sens)
        w = w * (1 + err / err.max())
        w /= w.max()
    return coef


def eval_f32(coef, y):
    t = np.abs(y).astype(np.float32)
    p = np.float32(coef[-1]) * np.ones_like(t)
    for c in coef[-2::-1]:
        p = (p * t + np.float32(c)).astype(np.float32)
    return (np.maximum(y, 0) - np.abs(y) * np.exp2(p).astype(np.float32)).astype(np.float32)


def erf_poly_f32(y):   # gelu_erf_s of csrc/encoder.hip
    CP = np.float32(4.52548360824585)
    yc = np.clip(y, -CP, CP).astype(np.float32)
    t = (yc * yc * np.float32(0.09765625) - np.float32(1)).astype(np.float32)
    K = [8.469007444e-04, -2.387454268e-03, 3.280109027e-03, -5.588355009e-03, 1.136882324e-02, -1.921003498e-02,
         2.861942165e-02, -4.021260887e-02, 5.456056446e-02, -7.682786137e-02, 1.560353935e-01]
    p = (t * np.float32(K[0]) + np.float32(K[1])).astype(np.float32)
    for k in K[2:]:
        p = (t * p + np.float32(k)).astype(np.float32)
    return (y * (yc * p + np.float32(0.5))).astype(np.float32)


if __name__ == "__main__":
    coef = fit()
    y = np.linspace(-40, 40, 2000001).astype(np.float32)
    ref = y.astype(np.float64) * ndtr(y.astype(np.float64))
    print("P coefficients (t^0 .. t^5):", [float(np.float32(c)) for c in coef])
    print("max |error|, 2^P form:        %.3e" % np.abs(eval_f32(coef, y) - ref).max())
    print("max |error|, erf polynomial:  %.3e" % np.abs(erf_poly_f32(y) - ref).max())

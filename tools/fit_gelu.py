"""Coefficients of `gelu_erf` (emma-x_amd/csrc/common.h): x Phi(x) = max(x, 0) - |x| h(|x|), h = erfc(|x| / sqrt 2) / 2 = exp2(Q(|x|)).

Q = log2(h) is fitted by a degree-6 polynomial on [0, 6] with Lawson-reweighted least squares, the weight being the sensitivity of
|x| h to an error in Q (|x| h ln 2): what is minimised is the absolute error of the GELU value, not of Q.  The script prints the
coefficients and the error of the fp32 evaluation (Horner, clamp at 6) against the float64 erf form on a dense grid, beside the
error of the form used in rounds 2-4 (Abramowitz-Stegun 7.1.26: rcp + exp2).  CPU only:  python tools/fit_gelu.py
"""
import numpy as np
from scipy.special import erf, erfc

A, DEG = 6.0, 6


def fit():
    a = (np.cos(np.pi * (np.arange(6000) + 0.5) / 6000) + 1) / 2 * A
    h = erfc(a / np.sqrt(2)) / 2
    tgt, w = np.log2(h), np.ones_like(a)
    for _ in range(200):
        sens = a * h * np.log(2) + 1e-12
        V = np.polynomial.polynomial.polyvander(a / A, DEG)
        coef, *_ = np.linalg.lstsq(V * (w * sens)[:, None], tgt * (w * sens), rcond=None)
        err = np.abs(a * (2.0 ** (V @ coef)) - a * h)
        w = w * (1 + 2 * err / err.max())
        w /= w.mean()
    return coef / A ** np.arange(DEG + 1), err.max()


def report(name, g, xs):
    ref = 0.5 * xs.astype(np.float64) * (1 + erf(xs.astype(np.float64) / np.sqrt(2)))
    e = np.abs(g - ref)
    big = np.abs(ref) > 1e-3
    print(f"{name}: max abs error {e.max():.3e} at x = {xs[e.argmax()]:.4f}; worst error / bf16 ulp of the result {(e[big] / (np.abs(ref[big]) * 2.0 ** -9)).max():.4f}")


if __name__ == "__main__":
    c, e64 = fit()
    print("float64 fit, max |error| of |x| h:", e64)
    print("coefficients c0 .. c6:", ", ".join(f"{x:.10e}" for x in c))
    xs = np.concatenate([np.linspace(-8, 8, 400001), [-1e4, -100.0, -25.0, 25.0, 100.0, 1e4]]).astype(np.float32)
    c32 = c.astype(np.float32)
    a = np.minimum(np.abs(xs), np.float32(A))
    q = np.full_like(a, c32[-1])
    for k in c32[-2::-1]:
        q = (q * a + k).astype(np.float32)
    report("exp2(Q) form, fp32", np.maximum(xs, 0) - np.abs(xs) * np.exp2(q).astype(np.float32), xs)
    zs = (np.abs(xs) * np.float32(0.84932180028801904272)).astype(np.float32)
    t = (np.float32(1) / (np.float32(0.2727374809) * zs + np.float32(1))).astype(np.float32)
    pl = np.float32(0.5307027145) * t + np.float32(-0.7265760135)
    for k in (0.7107068705, -0.142248368, 0.127414796):
        pl = (pl * t + np.float32(k)).astype(np.float32)
    report("A-S 7.1.26 form (rounds 2-4), fp32", np.maximum(xs, 0) - np.abs(xs) * (pl * t * np.exp2(-(zs * zs))).astype(np.float32), xs)

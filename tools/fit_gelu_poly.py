"""Polynomial forms of the erf-GELU pieces for the fast-mode GEMM epilogues (csrc/common.h: gelu_fast2 / gelu_grad_fast2).

    Phi(x)   = 0.5 + xc * R(t)      gelu(x)  = x * Phi(x)
    gelu'(x) = 0.5 + xc * S(t)      (= Phi(x) + x*phi(x))
with xc = clamp(x, -XMAX, XMAX), t = 2*xc^2/XMAX^2 - 1 in [-1, 1] and R, S evaluated by Horner in t (coefficients in t are O(0.1),
so f32 Horner is well conditioned; a polynomial in x^2 directly would lose ~4 digits to cancellation).  Chebyshev interpolation
of (f(x) - 0.5)/x in u = x^2.  No transcendental instruction, all FMAs pack two lanes-worth per v_pk_fma_f32.
Prints the C arrays and the f32-evaluated maximum errors; run:  python tools/fit_gelu_poly.py"""
import numpy as np
from numpy.polynomial import chebyshev as C
from scipy.special import erf


def Phi(x):
    return 0.5 * (1 + erf(x / np.sqrt(2)))


def dgelu(x):
    return Phi(x) + x * np.exp(-x * x / 2) / np.sqrt(2 * np.pi)


def fit(fun, xmax, deg):
    n = 8000
    t = np.cos(np.pi * (np.arange(n) + 0.5) / n)
    x = np.sqrt((t + 1) / 2 * xmax ** 2)
    return C.cheb2poly(C.chebfit(t, (fun(x) - 0.5) / x, deg))


def eval32(mono, xmax, x):
    x = x.astype(np.float32)
    xc = np.clip(x, -np.float32(xmax), np.float32(xmax))
    t = xc * xc * np.float32(2 / xmax ** 2) + np.float32(-1)
    r = np.full_like(t, np.float32(mono[-1]))
    for c in mono[-2::-1]:
        r = r * t + np.float32(c)
    return np.float32(0.5) + xc * r


if __name__ == "__main__":
    rng = np.random.RandomState(0)
    xs = np.concatenate([np.linspace(-9, 9, 600001), rng.randn(300000) * 1.5])
    # bf16 pass: (PHI 4.0, 8), (DGELU 4.5, 9); half pass (csrc/common.h, TAD_OPND_F16): (PHI 5.0, 12), (DGELU 5.0, 13)
    for name, fun, xmax, deg in (("PHI", Phi, 4.0, 8), ("DGELU", dgelu, 4.5, 9), ("PHI", Phi, 5.0, 12), ("DGELU", dgelu, 5.0, 13)):
        mono = fit(fun, xmax, deg)
        got = eval32(mono, xmax, xs).astype(np.float64)
        err = np.max(np.abs(got - fun(xs)))
        print(f"// {name}: XMAX {xmax}, degree {deg} in t; max |error| evaluated in f32 = {err:.2e}"
              + (f"; gelu = x*Phi: {np.max(np.abs(xs * got - xs * fun(xs))):.2e}" if name == "PHI" else ""))
        print(f"constexpr float {name}_XMAX = {xmax}f;")
        print(f"constexpr float {name}_C[{deg + 1}] = {{" + ", ".join(f"{c:.9e}f" for c in mono) + "};")

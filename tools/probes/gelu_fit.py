"""Coefficients of gelu_erf (csrc/s6d_gemm.hip): degree-7 minimax fit (Lawson-weighted Chebyshev least squares) of
P(z) = log2(erfc(z / sqrt 2) / 2) on [0, 6], rounded to float32, and its error as the kernel evaluates it (float32 Horner with
fused multiply-add, v_exp_f32), against round 2's Abramowitz-Stegun form.  Host only (numpy / scipy / torch)."""
import numpy as np
import torch
from numpy.polynomial import chebyshev as C
from scipy.special import erfc

f = np.float32
zmax, deg = 6.0, 7
z = np.linspace(0, zmax, 400001)
Q = 0.5 * erfc(z / np.sqrt(2))
L = np.log2(Q)
t = 2 * z / zmax - 1
w = np.ones_like(z)
for _ in range(80):
    c = C.chebfit(t, L, deg, w=w)
    e = np.abs(C.chebval(t, c) - L)
    w = w * (e / e.max() + 1e-3)
    w /= w.max()
coef = np.polynomial.Polynomial(C.cheb2poly(c))(np.polynomial.Polynomial([-1, 2 / zmax])).coef.astype(f)
print("P coefficients, constant first:", ", ".join("%.9ef" % v for v in coef))


def new(x):
    zz = np.minimum(np.abs(x), f(zmax))
    acc = np.full_like(zz, coef[-1])
    for a in coef[-2::-1]:
        acc = (acc.astype(np.float64) * zz + a).astype(f)
    return (np.maximum(x, 0).astype(np.float64) - np.abs(x).astype(np.float64) * np.exp2(acc).astype(f)).astype(f)


def round2(x):
    ax = np.abs(x)
    zs = (ax * f(0.84932180028801904)).astype(f)
    e = np.exp2((-zs * zs).astype(f)).astype(f)
    tt = (f(1) / ((f(0.3275911 * 0.70710678118654752) * ax + f(1)).astype(f))).astype(f)
    p = (tt * f(0.5 * 1.061405429) + f(0.5 * -1.453152027)).astype(f)
    for v in (0.5 * 1.421413741, 0.5 * -0.284496736, 0.5 * 0.254829592):
        p = (tt * p + f(v)).astype(f)
    return (-ax * (p * tt * e).astype(f) + np.maximum(x, f(0))).astype(f)


x = (np.random.default_rng(0).standard_normal(4_000_000) * 2.0).astype(f)
ref = torch.nn.functional.gelu(torch.from_numpy(x).double())
exact = ref.to(torch.bfloat16).float()
for nm, fn in (("round-2 form (A&S 7.1.26)", round2), ("2^P(|x|) form", new)):
    g = torch.from_numpy(fn(x))
    rel = (g.double() - ref).abs() / (ref.abs() + 1e-12)
    near = torch.from_numpy(np.abs(x) < 5)
    print(f"{nm}: max relative error for |x| < 5: {rel[near].max().item():.2e}; bf16 results that differ from the correctly rounded "
          f"ones: {100 * (g.to(torch.bfloat16).float() != exact).float().mean().item():.3f} %")

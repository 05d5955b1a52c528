"""Coefficients of gelu_erf_fast (csrc/common.h): log2 Q(a), Q(a) = erfc(a / sqrt 2) / 2, on [0, 6.5] by a degree-6 polynomial in a
(Chebyshev-node interpolation, i.e. near-minimax), and the resulting error of gelu(x) = max(x, 0) - |x| Q(|x|).   python tools_dev/fit_gelu.py"""
import numpy as np
from numpy.polynomial import chebyshev as C
from scipy.special import erfc

XM = 6.5
xs = np.cos(np.pi * (np.arange(4000) + 0.5) / 4000) * XM / 2 + XM / 2
xt = np.linspace(0, XM, 200001)
for deg in (5, 6, 7):
    p = C.Chebyshev.fit(xs, np.log2(0.5 * erfc(xs / np.sqrt(2))), deg, domain=[0, XM]).convert(kind=np.polynomial.Polynomial)
    q, qt = np.exp2(p(xt)), 0.5 * erfc(xt / np.sqrt(2))
    print(deg, "rel err of Q %.2e" % np.abs(q / qt - 1).max(), " abs err of gelu %.2e" % np.abs(xt * (q - qt)).max())
    print("   ", ", ".join("%.10ef" % v for v in p.coef))


# dgelu_erf_fast2: gelu'(x) = 1/2 + sign(x) (1/2 - e U(a)), e = exp(-a^2/2), U = Q / e - a / sqrt(2 pi); degree-6 fit of U weighted by e
# (iteratively re-weighted least squares towards the minimax error of e U)
a = np.linspace(0, XM, 200001)
e = np.exp(-a * a / 2)
U = 0.5 * erfc(a / np.sqrt(2)) / e - a / np.sqrt(2 * np.pi)
for deg in (5, 6, 7):
    w = e.copy()
    for _ in range(60):
        V = np.vander(a, deg + 1, increasing=True)
        coef = np.linalg.lstsq(V * w[:, None], U * w, rcond=None)[0]
        err = (V @ coef - U) * e
        w = w * (1 + 4 * np.abs(err) / np.abs(err).max())
        w /= w.max()
    print(deg, "abs err of gelu' %.2e" % np.abs(err).max())
    print("   ", ", ".join("%.10ef" % v for v in coef))

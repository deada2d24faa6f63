"""Fit the polynomial coefficients used by include/glabc_numerics.h.

Absolute-error least squares on Chebyshev nodes in float64 (no division by
small powers, so the fits are well conditioned); coefficients are then rounded
to float32.  Run by hand; the printed numbers are pasted into the header and
their accuracy is measured exhaustively by tests/test_numerics.py.
Not used at run time.
"""
import numpy as np

def cheb_nodes(a, b, n):
    k = np.arange(n)
    x = np.cos(np.pi * (2 * k + 1) / (2 * n))
    return 0.5 * (a + b) + 0.5 * (b - a) * x

def fit(y, x, powers, rel=None):
    A = np.stack([x ** p for p in powers], axis=1)
    if rel is not None:           # minimise relative error: divide rows by |reference value|
        A = A / rel[:, None]
        y = y / rel
    c, *_ = np.linalg.lstsq(A, y, rcond=None)
    return c

def show(name, c):
    print(name)
    for v in c:
        print("   %-22s /* %r */" % (float(np.float32(v)).hex(), float(np.float32(v))))

n = 20000
# f32 log: log(1+f) = f - f^2/2 + f^3 P(f)
f = cheb_nodes(np.sqrt(0.5) - 1, np.sqrt(2.0) - 1, n)
y = np.log1p(f) - f + 0.5 * f * f
c = fit(y, f, range(3, 11), rel=np.maximum(np.abs(np.log1p(f)), 1e-3))
show("logf: coefficients of f^3..f^10", c)

# f32 exp: exp(r) = 1 + r + r^2 Q(r)
h = np.log(2) / 2 * 1.0001
r = cheb_nodes(-h, h, n)
y = np.expm1(r) - r
c = fit(y, r, range(2, 7), rel=np.exp(r))
show("expf: coefficients of r^2..r^6", c)

# sin/cos on [-pi/4, pi/4]
m = np.pi / 4 * 1.0001
a = cheb_nodes(-m, m, n)
y = np.sin(a) - a
c = fit(y, a, (3, 5, 7), rel=np.maximum(np.abs(np.sin(a)), 1e-3))
show("sin: coefficients of a^3,a^5,a^7", c)
y = -2 * np.sin(a / 2) ** 2 + 0.5 * a * a      # cos a - 1 + a^2/2 without cancellation in cos
c = fit(y, a, (4, 6, 8))
show("cos: coefficients of a^4,a^6,a^8", c)

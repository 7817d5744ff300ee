import numpy as np
from scipy.special import erf
from numpy.polynomial import chebyshev as C

def fit_odd(f, c, n, iters=30):
    """minimax-ish (Remez via iterated weighted LSQ / Lawson) fit f(u) ~ u*P(u^2), deg P = n, on [0,c]"""
    u = np.cos(np.linspace(0, np.pi, 4001))*0.5+0.5
    u = u*c
    u = u[u>1e-9]
    A = np.stack([u**(2*k+1) for k in range(n+1)],1)
    y = f(u)
    w = np.ones_like(u)
    for _ in range(iters):
        coef, *_ = np.linalg.lstsq(A*w[:,None], y*w, rcond=None)
        err = np.abs(A@coef-y)
        w = w*(0.5+err/err.max())   # Lawson-style
        w/=w.max()
    return coef, np.abs(A@coef-y).max()

def horner32(coef, x, c):
    x = x.astype(np.float32)
    u = np.clip(x, -np.float32(c), np.float32(c))
    s = u*u
    p = np.float32(coef[-1])*np.ones_like(s)
    for k in range(len(coef)-2, -1, -1):
        p = p*s+np.float32(coef[k])
    return u*p

Phi = lambda x: 0.5*(1+erf(x/np.sqrt(2)))
phi = lambda x: np.exp(-x*x/2)/np.sqrt(2*np.pi)
x = np.linspace(-8, 8, 400001)
for c in (3.5, 3.8, 4.0, 4.2, 4.5):
    for n in (4,5,6,7):
        co, e = fit_odd(lambda u: 0.5*erf(u/np.sqrt(2)), c, n)
        ph = 0.5+horner32(co, x, c).astype(np.float64)
        g = x*ph
        eg = np.abs(g-x*Phi(x)).max()
        ephi = np.abs(ph-Phi(x)).max()
        print(f"Phi c={c} n={n}: fit {e:.2e} errPhi {ephi:.2e} errgelu {eg:.2e}")
print()
for c in (4.0, 4.5, 5.0):
    for n in (5,6,7,8):
        co, e = fit_odd(lambda u: 0.5*erf(u/np.sqrt(2))+u*phi(u), c, n)
        gd = 0.5+horner32(co, x, c).astype(np.float64)
        ed = np.abs(gd-(Phi(x)+x*phi(x))).max()
        print(f"dgelu c={c} n={n}: fit {e:.2e} err {ed:.2e}")

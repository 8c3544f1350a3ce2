#!/usr/bin/env python3
"""Minimax fit of sin(r) = r + r^3 Q(r^2) on |r| <= pi/2 + 0.02 (relative error), Q of degree `deg`, and a Float32
emulation of the device's cos/sin (magic-add reduction, FMA Cody-Waite, Horner) measuring its worst error in ulp.

    python tools/fit/trig_fit.py [deg]

The coefficients in csrc/de_device_ops.h (DE_TRIG_S*) come from here."""
import sys
import numpy as np

def remez_rel(deg, R, iters=40):
    # approximate g(z) = (sin(r)/r - 1)/z, z = r^2, by Q(z) of degree deg minimising |r + r^3 Q - sin r| / |sin r|
    # = | z (Q - g) | / (sin(r)/r): weighted Remez on z in [0, R^2] with weight w(z) = z / (sin r / r)
    Z = R * R
    n = deg + 2
    k = np.arange(n)
    z = Z * 0.5 * (1 - np.cos(np.pi * k / (n - 1)))
    z[0] = Z * 1e-6
    def g(z):
        r = np.sqrt(z)
        r = np.asarray(r, dtype=np.longdouble)
        zz = r * r
        # series for small z to avoid cancellation
        s = np.where(zz < 1e-2, -1/6 + zz/120 - zz*zz/5040 + zz**3/362880, (np.sin(r)/r - 1)/np.where(zz == 0, 1, zz))
        return s.astype(np.float64)
    def w(z):
        r = np.sqrt(z)
        return z / (np.sin(r) / r)
    for _ in range(iters):
        A = np.zeros((n, n))
        for j in range(deg + 1):
            A[:, j] = z ** j
        A[:, deg + 1] = (-1.0) ** k / w(z)
        sol = np.linalg.solve(A, g(z))
        c, E = sol[:deg + 1], sol[deg + 1]
        zz = np.linspace(Z * 1e-6, Z, 200001)
        err = (np.polyval(c[::-1], zz) - g(zz)) * w(zz)
        # new extrema: split at sign changes
        idx = [0]
        sgn = np.sign(err)
        ext = []
        start = 0
        for i in range(1, len(zz) + 1):
            if i == len(zz) or sgn[i] != sgn[start]:
                seg = slice(start, i)
                j = start + np.argmax(np.abs(err[seg]))
                ext.append(j)
                start = i
        if len(ext) < n:
            break
        # keep n largest alternating
        while len(ext) > n:
            # drop the smaller of the end points
            if abs(err[ext[0]]) < abs(err[ext[-1]]):
                ext.pop(0)
            else:
                ext.pop()
        z = zz[ext]
    return c, np.max(np.abs(err))

def f32(x):
    return np.asarray(x, dtype=np.float32)
def fma32(a, b, c):
    return f32(a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64))

def emulate(x, S, sin=False, terms=3):
    x = f32(x)
    INV_PI = np.float32(float.fromhex('0x1.45f306p-2'))
    MAGIC = np.float32(12582912.0)
    P = [np.float32(float.fromhex(h)) for h in ('0x1.921fb4p+1', '0x1.4442d0p-23', '0x1.846988p-47')]  # pi rounded DOWN term by term: all positive (sign of zero)
    t = f32(x * INV_PI) if sin else fma32(x, np.full_like(x, INV_PI), np.full_like(x, np.float32(0.5)))
    kk = f32(t + MAGIC)
    n = f32(kk - MAGIC)
    m = n if sin else f32(n - np.float32(0.5))
    r = x
    for p in P[:terms]:
        r = fma32(-m, np.full_like(x, p), r)
    z = f32(r * r)
    S = [np.float32(s) for s in S]
    p = np.full_like(x, S[-1])
    for s in S[-2::-1]:
        p = fma32(z, p, np.full_like(x, s))
    s_ = fma32(f32(r * z), p, r)
    par = (kk.view(np.uint32) & 1).astype(bool)
    return np.where(par, -s_, s_), r

def ulp_err(y, ref):
    y = y.astype(np.float64)
    u = np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
    return np.abs(y - ref) / u

if __name__ == "__main__":
    deg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    c, e = remez_rel(deg, np.pi / 2 + 0.02)
    print("deg", deg, "minimax rel err", e)
    c32 = [np.float32(v) for v in c]
    print("coeffs:", [float(v).hex() for v in c32], [float(v) for v in c32])
    rng = np.random.default_rng(1)
    for lim in (4.0, 100.0, 1e4, 1e5):
        x = np.concatenate([rng.uniform(-lim, lim, 4_000_000), (np.arange(-2000, 2000) * (np.pi / 2)), np.linspace(-lim, lim, 1_000_001)]).astype(np.float32)
        # also floats adjacent to multiples of pi/2
        near = f32(np.arange(1, int(lim / (np.pi / 2))) [:200000] * (np.pi / 2))
        x = np.concatenate([x, near, np.nextafter(near, np.float32(np.inf)), np.nextafter(near, np.float32(-np.inf))])
        for terms in (3, 2):
            y, r = emulate(x, c32, False, terms)
            ref = np.cos(x.astype(np.float64))
            u = ulp_err(y, ref)
            i = int(np.argmax(u))
            print(f"|x|<={lim:g} terms={terms}: cos max ulp {u.max():.3f} at x={x[i]!r} r={r[i]!r}; >1.0: {(u > 1.0).mean():.2e}; max |y| {np.abs(y).max()!r}")

def with_fix(x, S, sin=False):
    y, r = emulate(x, S, sin, 3)
    near = np.abs(np.abs(r) - np.float32(float.fromhex('0x1.921fb6p+0'))) < np.float32(float.fromhex('0x1.fep-13'))
    one = np.copysign(np.float32(1), r)
    par = (f32(f32((x * np.float32(float.fromhex('0x1.45f306p-2'))) if sin else fma32(x, np.full_like(x, np.float32(float.fromhex('0x1.45f306p-2'))), np.full_like(x, np.float32(0.5)))) + np.float32(12582912.0)).view(np.uint32) & 1).astype(bool)
    return np.where(near, np.where(par, -one, one), y), r, near

import struct, math, random
import mpmath
mpmath.mp.prec = 200

def bits(x): return struct.unpack('<q', struct.pack('<d', x))[0]
def frombits(b): return struct.unpack('<d', struct.pack('<q', b))[0]
HI = ~0xFFFFFFFF
def trunc32(v): return frombits(bits(v) & HI)

def two_sum(a, b):
    s = a + b; bb = s - a
    return s, (a - (s - bb)) + (b - bb)
def fast_two_sum(a, b):
    s = a + b
    return s, b - (s - a)
def split(a):
    c = 134217729.0 * a
    h = c - (c - a)
    return h, a - h
def two_prod(a, b):
    p = a * b
    ah, al = split(a); bh, bl = split(b)
    return p, ((ah * bh - p) + ah * bl + al * bh) + al * bl
def dd_mul(ah, al, bh, bl):
    p, e = two_prod(ah, bh)
    e += ah * bl + al * bh
    return fast_two_sum(p, e)
def dd_add(ah, al, bh, bl):
    s, e = two_sum(ah, bh)
    e += al + bl
    return fast_two_sum(s, e)

def log2_core(ax):
    """log2(ax) = t1 + t2, ax > 0 finite (msun pow's log stage)"""
    n = 0
    b = bits(ax)
    if (b >> 52) == 0:
        ax *= 2.0**53; n = -53; b = bits(ax)
    ix = (b >> 32) & 0xFFFFFFFF
    n += (ix >> 20) - 0x3ff
    j = ix & 0x000fffff
    ix = j | 0x3ff00000
    if j <= 0x3988E: k = 0
    elif j < 0xBB67A: k = 1
    else:
        k = 0; n += 1; ix -= 0x00100000
    ax = frombits((ix << 32) | (b & 0xFFFFFFFF))
    bp = 1.5 if k else 1.0
    dp_h = 5.84962487220764160156e-01 if k else 0.0
    dp_l = 1.35003920212974897128e-08 if k else 0.0
    u = ax - bp; v = 1.0 / (ax + bp)
    ss = u * v
    s_h = trunc32(ss)
    t_h = frombits(((((ix >> 1) | 0x20000000) + 0x00080000 + (k << 18)) & 0xFFFFFFFF) << 32)
    t_l = ax - (t_h - bp)
    s_l = v * ((u - s_h * t_h) - s_h * t_l)
    s2 = ss * ss
    r = s2 * s2 * (5.99999999999994648725e-01 + s2 * (4.28571428578550184252e-01 + s2 * (3.33333329818377432918e-01 +
        s2 * (2.72728123808534006489e-01 + s2 * (2.30660745775561754067e-01 + s2 * 2.06975017800338417784e-01)))))
    r += s_l * (s_h + ss)
    s2 = s_h * s_h
    t_h = trunc32(3.0 + s2 + r)
    t_l = r - ((t_h - 3.0) - s2)
    u = s_h * t_h
    v = s_l * t_h + t_l * ss
    p_h = trunc32(u + v)
    p_l = v - (p_h - u)
    z_h = 9.61796700954437255859e-01 * p_h
    z_l = -7.02846165095275826516e-09 * p_h + p_l * 9.61796693925975554329e-01 + dp_l
    t = float(n)
    t1 = trunc32(((z_h + z_l) + dp_h) + t)
    t2 = z_l - (((t1 - t) - dp_h) - z_h)
    return t1, t2

def exp2_core(p_h, p_l):
    z = p_l + p_h
    if z >= 1024.0: return math.inf
    if z <= -1075.0: return 0.0
    ni = 0
    if abs(z) > 0.5:
        zi = float(round(z))  # rint: ties to even; python round is ties-to-even
        ni = int(zi); p_h -= zi
    t = trunc32(p_l + p_h)
    u = t * 6.93147182464599609375e-01
    v = (p_l - (t - p_h)) * 6.93147180559945286227e-01 + t * -1.90465429995776804525e-09
    z = u + v
    w = v - (z - u)
    t = z * z
    tt = z - t * (1.66666666666666019037e-01 + t * (-2.77777777770155933842e-03 + t * (6.61375632143793436117e-05 +
         t * (-1.65339022054652515390e-06 + t * 4.13813679705723846039e-08))))
    r = (z * tt) / (tt - 2.0) - (w + z * w)
    z = 1.0 - (r - z)
    return math.ldexp(z, ni)

def dd_const(v):
    h = float(v); l = float(v - mpmath.mpf(h)); return h, l
LOG2E = dd_const(1 / mpmath.log(2))
HALF_LOG2_2PI = dd_const(mpmath.log(2 * mpmath.pi, 2) / 2)
INV_LN2 = LOG2E[0]
SC = [1/12., -1/360., 1/1260., -1/1680., 1/1188., -691/360360., 1/156., -3617/122400.]

def gamma_f64(x):
    if x != x or math.isinf(x) or x == 0.0 or (x < 0 and x == math.floor(x)) or x < -180.0:
        return None  # delegated
    if abs(x) < 2.0**-55: return 1.0 / x
    # shift up to y >= 16
    n = 0 if x >= 16.0 else int(math.ceil(16.0 - x))
    ph, pl, esum = 1.0, 0.0, 0
    yh, yl = x, 0.0
    for k in range(n):
        fh, fl = two_sum(x, float(k))
        ph, pl = dd_mul(ph, pl, fh, fl)
        if abs(ph) > 2.0**500:
            ph *= 2.0**-500; pl *= 2.0**-500; esum += 500
        elif abs(ph) < 2.0**-500:
            ph *= 2.0**500; pl *= 2.0**500; esum -= 500
    if n: yh, yl = two_sum(x, float(n))
    # F(y)
    t1, t2 = log2_core(yh)
    a_h = yh - 0.5
    a1 = trunc32(a_h)
    p_l = (a_h - a1) * t1 + a_h * t2 + yl * (t1 + t2) + a_h * (yl / yh) * INV_LN2  # (log2(yh + yl) = log2 yh + yl / (yh ln 2))
    p_h = a1 * t1
    eh, el = fast_two_sum(p_h, p_l)
    # - y log2 e
    qh, ql = two_prod(yh, LOG2E[0])
    ql += yh * LOG2E[1] + yl * LOG2E[0]
    qh, ql = fast_two_sum(qh, ql)
    eh, el = dd_add(eh, el, -qh, -ql)
    eh, el = dd_add(eh, el, HALF_LOG2_2PI[0], HALF_LOG2_2PI[1])
    # Stirling tail, y >= 16; derivative term of y_lo: d/dy of the tail is negligible
    r = 1.0 / yh; r2 = r * r
    s = SC[7]
    for c in SC[6::-1]: s = s * r2 + c
    s *= r
    el += s * INV_LN2
    # - log2|P|
    sign = 1.0
    if n:
        if ph < 0: sign = -1.0; ph, pl = -ph, -pl
        l1, l2 = log2_core(ph)
        l2 += (pl / ph) * INV_LN2
        eh, el = dd_add(eh, el, -l1 - float(esum), -l2)
    return sign * exp2_core(eh, el)

def ulp_err(x):
    g = gamma_f64(x)
    if g is None: return None
    ref = mpmath.gamma(mpmath.mpf(x))
    if math.isinf(g) or g == 0.0:
        return 0.0 if (abs(ref) > mpmath.mpf(2)**1024 or abs(ref) < mpmath.mpf(2)**-1075) else 99.0
    r = float(ref)
    u = math.ulp(r) if r != 0 else 5e-324
    return float(abs(mpmath.mpf(g) - ref) / u)

random.seed(1)
worst = (0, None)
pts = [random.uniform(0.05, 30) for _ in range(4000)] + [random.uniform(-5.9, -0.1) for _ in range(3000)] + \
      [random.uniform(30, 171.6) for _ in range(2000)] + [random.uniform(-170, -6) for _ in range(2000)] + \
      [10.0**random.uniform(-300, -1) for _ in range(500)] + [-(10.0**random.uniform(-300, -1)) for _ in range(500)] + \
      [float(k) for k in range(1, 172)] + [k + 0.5 for k in range(-170, 171)] + [-k + s * 2.0**-e for k in range(0, 30) for e in (10, 30, 50) for s in (1, -1)]
import collections
hist = collections.Counter()
for x in pts:
    e = ulp_err(x)
    if e is None: continue
    hist[min(int(e * 10), 20)] += 1
    if e > worst[0]: worst = (e, x)
print("worst", worst)
print(sorted(hist.items()))

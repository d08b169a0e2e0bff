#!/usr/bin/env python3
"""Generate the constants of the kernels' own FP64 elementary functions (csrc/afiv_math.h):
polynomial coefficients and reduction tables, 60-digit arithmetic (mpmath), each rounded once to
binary64; binary64 models of the table-driven functions are checked against mpmath below.

  exp r  = sum_{k<=13} r^k / k!      |r| <= ln(2)/2  (plain Taylor: remainder 4e-18)

Table-driven exp (afivm::exp_tab_f64): x = (128 m + j) ln(2)/128 + r, |r| <= ln(2)/256, table of
2^(j/128) as (hi, lo), exp r - 1 by the Taylor polynomial to r^5 (remainder r^6/720 = 5e-19, i.e.
2e-16 of exp r - 1 itself at the interval's ends, where exp x - 1 is returned as that polynomial).

Table-driven sincos (afivm::sincos_tab_f64): x = j pi/32 + r, |r| <= pi/64, table of
(sin, cos)(j pi/32), j < 64, correctly rounded; sin r and cos r - 1 by Taylor polynomials
(remainders r^9/9! = 5e-18 and r^10/10! = 2e-20).


Table-driven log1p (afivm::log1p_tab_f64): 1 + f = 2^k z with the high word of z in
[LOGT_OFF_HI, LOGT_OFF_HI + 2^20), 128 bins of 2^13 high-word values each, laid out so that 1.0 is
the middle of bin 80, whose entry is (1, 0) exactly: there r = z - 1 + c = f and the result keeps
its relative accuracy as f -> 0.  Other bins: c_i = bin midpoint, invc_i = a double within 4e-9 of 1/c_i picked so that
logc_i = -log invc_i is a double to 1e-4 ulp; r = z invc_i - 1 (one fma), |r| <= 2^-8, log1p r by Taylor to r^7.

Table-driven arctangent (afivm::atan2_tab_f64): t = min/max in [0, 1], j = rint(64 t), h = t - j/64,
atan t = sum_{k<8} A[j][k] h^k with |h| <= 1/128.  Row 0 is the odd Taylor series (0, 1, 0, -1/3,
0, 1/5, 0, -1/7: exact 0 and 1, so the result keeps its relative accuracy as t -> 0); rows 1..64
are degree-7 Chebyshev interpolants of atan about j/64 (error 1e-20).

Writes csrc/afiv_math_coeffs.inc.
"""
import os

import mpmath as mp
import numpy as np

mp.mp.dps = 60


def cheb_fit(fun, lo, hi, deg):
    """Monomial coefficients (in the original variable) of the degree-`deg` Chebyshev interpolant."""
    n = deg + 1
    nodes = [(lo + hi) / 2 + (hi - lo) / 2 * mp.cos(mp.pi * (2 * k + 1) / (2 * n)) for k in range(n)]
    A = mp.matrix(n, n)
    for i, x in enumerate(nodes):
        for j in range(n):
            A[i, j] = x ** j
    c = mp.lu_solve(A, mp.matrix([fun(x) for x in nodes]))
    return [c[j] for j in range(n)]


LOGT_OFF_HI = 0x3fe5f000


def log_table():
    import struct
    rows = []
    for i in range(128):
        lo = struct.unpack("<d", struct.pack("<Q", (LOGT_OFF_HI + i * 0x2000) << 32))[0]
        hi = struct.unpack("<d", struct.pack("<Q", (LOGT_OFF_HI + (i + 1) * 0x2000) << 32))[0]
        if lo < 1.0 < hi:
            rows.append((1.0, 0.0))
            continue
        c = (mp.mpf(lo) + mp.mpf(hi)) / 2
        # among the doubles next to 1/c take the one whose logarithm is closest to a double:
        # the entry's rounding error then drops from half an ulp to about 1e-4 ulp
        base = float(1 / c)
        best = None
        jr = np.random.default_rng(i)
        for j in [0] + [int(v) for v in jr.integers(-20_000_000, 20_000_000, 3000)]:
            # (offsets of up to 2e7 ulp, 4e-9 relative: a bin midpoint such as 255/256 moves its
            # logarithm by a whole number of ulps per ulp, so nearby candidates all round alike)
            cand = float(base + j * (np.nextafter(base, np.inf) - base))
            lg = -mp.log(mp.mpf(cand))
            err = abs((mp.mpf(float(lg)) - lg) / lg)
            if best is None or err < best[0]:
                best = (err, cand, float(lg))
        rows.append((best[1], best[2]))
    return rows


def log1p_tab_model(f, rows, ln2hi, ln2lo):
    """binary64 model of afivm::log1p_tab_f64 (np.float64 arithmetic, fma emulated in mpmath)."""
    import struct

    def fma(a, b, c):
        return float(mp.mpf(a) * mp.mpf(b) + mp.mpf(c))
    f = float(f)
    u = 1.0 + f
    c = f - (u - 1.0)
    bits = struct.unpack("<Q", struct.pack("<d", u))[0]
    hi = bits >> 32
    tmp = hi - LOGT_OFF_HI
    k = tmp >> 20                       # Python's >> floors: arithmetic shift
    i = (tmp >> 13) & 127
    zhi = hi - (k << 20)
    z = struct.unpack("<d", struct.pack("<Q", (zhi << 32) | (bits & 0xffffffff)))[0]
    invc, logc = rows[i]
    r = fma(z, invc, -1.0)
    r = fma(float(np.ldexp(c, -k)), invc, r)
    p = 1.0 / 7
    for cf in (-1.0 / 6, 1.0 / 5, -0.25, 1.0 / 3, -0.5):
        p = fma(p, r, cf)
    lg = fma(r * r, p, r)
    kd = float(k)
    return fma(kd, ln2hi, logc) + fma(kd, ln2lo, lg)


def atan_table():
    rows = [[0, 1, 0, -mp.mpf(1) / 3, 0, mp.mpf(1) / 5, 0, -mp.mpf(1) / 7]]
    a = mp.mpf(1) / 128
    for j in range(1, 65):
        tj = mp.mpf(j) / 64
        rows.append(cheb_fit(lambda h: mp.atan(tj + h), -a * mp.mpf("1.01"), a * mp.mpf("1.01"), 7))
    return [[float(v) for v in r] for r in rows]


def atan_tab_model(t, rows):
    def fma(a, b, c):
        return float(mp.mpf(a) * mp.mpf(b) + mp.mpf(c))
    j = int(np.rint(t * 64.0))
    h = fma(float(j), -1.0 / 64, t)
    c = rows[j]
    p = c[7]
    for k in range(6, -1, -1):
        p = fma(p, h, c[k])
    return p


def exp_table():
    """(hi, lo) of 2^(j/128), j < 128: hi the nearest double, lo the remainder."""
    rows = []
    for j in range(128):
        v = mp.mpf(2) ** (mp.mpf(j) / 128)
        hi = float(v)
        rows.append((hi, float(v - mp.mpf(hi))))
    return rows


def ln2_over_128_split():
    """ln(2)/128 = hi + lo with hi carrying 32 significant bits, so that k hi is exact for |k| < 2^21."""
    import struct
    v = mp.log(2) / 128
    bits = struct.unpack("<Q", struct.pack("<d", float(v)))[0]
    hi = struct.unpack("<d", struct.pack("<Q", bits & ~((1 << 21) - 1)))[0]
    return hi, float(v - mp.mpf(hi))


def exp_tab_model(x, rows, hi, lo, inv):
    """binary64 model of afivm::exp_tab_f64 / exp_expm1_tab_f64: returns (exp x, exp x - 1)."""
    def fma(a, b, c):
        return float(mp.mpf(a) * mp.mpf(b) + mp.mpf(c))
    k = float(np.rint(x * inv))
    r = fma(k, -hi, x)
    r = fma(k, -lo, r)
    ki = int(k)
    j, m = ki & 127, ki >> 7
    p = 1.0 / 120
    for cf in (1.0 / 24, 1.0 / 6, 0.5):
        p = fma(p, r, cf)
    q = fma(p, r, 1.0)
    w = q * r                                    # exp(r) - 1
    th, tl = rows[j]
    e = float(np.ldexp(fma(th, w, th), m))       # the plain exponential uses the high part only
    v = fma(th, w, tl)
    s = float(np.ldexp(1.0, m))
    ts, vs = th * s, v * s
    return e, ts + vs, (ts - 1.0) + vs


def horner(c, z):
    acc = np.float64(c[-1])
    for v in c[-2::-1]:
        acc = acc * z + np.float64(v)
    return acc


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    E = [1 / mp.factorial(k) for k in range(14)]

    rng = np.random.default_rng(1)
    # ---- float64 accuracy checks against mpmath ------------------------------------------------
    Ed = [float(v) for v in E]
    worst = {"exp": 0}
    for r in rng.uniform(-np.log(2) / 2, np.log(2) / 2, 4000):
        v = horner(Ed, r)
        ref = mp.exp(mp.mpf(r))
        worst["exp"] = max(worst["exp"], abs(float((mp.mpf(v) - ref) / ref)))
    rows = log_table()
    ln2 = mp.log(2)
    l1 = float(ln2)
    l2 = float(ln2 - mp.mpf(l1))
    worst["log1p_tab"] = 0
    fs = np.concatenate([rng.uniform(-0.95, 8, 6000), 10.0 ** rng.uniform(-20, 6, 2000), -(10.0 ** rng.uniform(-20, -0.1, 2000)),
                         [2.0 ** -9 * s for s in (-1, -0.999, -1.001)], [2.0 ** -8 * s for s in (1, 0.999, 1.001)],
                         rng.uniform(-0.01, 0.01, 3000)])
    for f in fs:
        ref = mp.log1p(mp.mpf(float(f)))
        v = log1p_tab_model(f, rows, l1, l2)
        worst["log1p_tab"] = max(worst["log1p_tab"], abs(float((mp.mpf(v) - ref) / ref)))
    arows = atan_table()
    worst["atan_tab"] = 0
    for t in np.concatenate([rng.uniform(0, 1, 8000), 10.0 ** rng.uniform(-15, -1, 1500), [1.0, 1.0 / 128, 3.0 / 128, 0.5]]):
        ref = mp.atan(mp.mpf(float(t)))
        worst["atan_tab"] = max(worst["atan_tab"], abs(float((mp.mpf(atan_tab_model(float(t), arows)) - ref) / ref)))
    erows = exp_table()
    e_hi, e_lo = ln2_over_128_split()
    e_inv = float(128 / mp.log(2))
    worst["exp_tab"] = worst["exp_tab2"] = worst["expm1_tab"] = 0
    xs = np.concatenate([rng.uniform(-700, 700, 3000), rng.uniform(-2, 2, 4000), rng.uniform(-0.05, 0.05, 3000),
                         10.0 ** rng.uniform(-300, -1, 1500), -(10.0 ** rng.uniform(-300, -1, 1500))])
    for x in xs:
        x = float(x)
        e, e2, em1 = exp_tab_model(x, erows, e_hi, e_lo, e_inv)
        ref = mp.exp(mp.mpf(x))
        worst["exp_tab"] = max(worst["exp_tab"], abs(float((mp.mpf(e) - ref) / ref)))
        worst["exp_tab2"] = max(worst["exp_tab2"], abs(float((mp.mpf(e2) - ref) / ref)))
        refm = mp.expm1(mp.mpf(x))
        worst["expm1_tab"] = max(worst["expm1_tab"], abs(float((mp.mpf(em1) - refm) / refm)))
    print("binary64 accuracy (relative):")
    for k, v in worst.items():
        print(f"  {k:9s} {v:.3e}")

    out = os.path.join(root, "arbitrage-free-iv-surfaces-using-vae_amd", "csrc", "afiv_math_coeffs.inc")
    with open(out, "w") as fh:
        fh.write("/* Generated by tools/gen_math_coeffs.py -- do not edit.\n"
                 " * Constants of the FP64 elementary functions in afiv_math.h (binary64 hex).\n"
                 " * Measured binary64 accuracy (relative): exp %.1e on its reduced interval, log1p table %.1e,\n"
                 " * atan table %.1e. */\n" % (worst["exp"], worst["log1p_tab"], worst["atan_tab"]))
        for name, vals in (("AFIV_EXP_C", E),):
            fh.write("#define %s_N %d\n" % (name, len(vals)))
            for i, v in enumerate(vals):
                fh.write("#define %s%d %s\n" % (name, i, float(v).hex()))
        ln2 = mp.log(2)
        l1 = float(ln2)
        l2 = float(ln2 - mp.mpf(l1))
        for name, v in (("AFIV_LN2_HI", l1), ("AFIV_LN2_LO", l2), ("AFIV_LOG2E", float(1 / ln2)),
                        ("AFIV_PI", float(mp.pi)), ("AFIV_PIO4", float(mp.pi / 4))):
            fh.write("#define %s %s\n" % (name, float(v).hex()))
        # table-driven exp: ln(2)/128 in two pieces, 128/ln(2), (hi, lo) of 2^(j/128)
        for name, v in (("AFIV_LN2O128_HI", e_hi), ("AFIV_LN2O128_LO", e_lo), ("AFIV_128_OVER_LN2", e_inv)):
            fh.write("#define %s %s\n" % (name, float(v).hex()))
        fh.write("#define AFIV_EXP_TAB_LIST \\\n")
        fh.write(", \\\n".join("  %s, %s" % (a.hex(), b.hex()) for a, b in erows) + "\n")
        # table-driven sincos: pi/32 in two pieces, Taylor coefficients, (sin, cos)(j pi/32)
        pio32 = mp.pi / 32
        q1 = float(pio32)
        q2 = float(pio32 - mp.mpf(q1))
        for name, v in (("AFIV_PIO32_1", q1), ("AFIV_PIO32_2", q2), ("AFIV_32_OVER_PI", float(32 / mp.pi)),
                        ("AFIV_SINT_C0", float(-mp.mpf(1) / 6)), ("AFIV_SINT_C1", float(mp.mpf(1) / 120)),
                        ("AFIV_SINT_C2", float(-mp.mpf(1) / 5040)),
                        ("AFIV_COST_C0", float(mp.mpf(1) / 24)), ("AFIV_COST_C1", float(-mp.mpf(1) / 720)),
                        ("AFIV_COST_C2", float(mp.mpf(1) / 40320))):
            fh.write("#define %s %s\n" % (name, float(v).hex()))
        fh.write("#define AFIV_ATAN_TAB_LIST \\\n")
        fh.write(", \\\n".join("  " + ", ".join(v.hex() for v in r) for r in arows) + "\n")
        fh.write("#define AFIV_LOGT_OFF_HI 0x%x\n" % LOGT_OFF_HI)
        for i, v in enumerate((-0.5, mp.mpf(1) / 3, -0.25, mp.mpf(1) / 5, -mp.mpf(1) / 6, mp.mpf(1) / 7)):
            fh.write("#define AFIV_LOGT_C%d %s\n" % (i, float(v).hex()))
        fh.write("#define AFIV_LOG_TAB_LIST \\\n")
        fh.write(", \\\n".join("  %s, %s" % (a.hex(), b.hex()) for a, b in rows) + "\n")
        fh.write("#define AFIV_SINCOS_TAB_LIST \\\n")
        rows = ["  %s, %s" % (float(mp.sin(j * pio32)).hex(), float(mp.cos(j * pio32)).hex()) for j in range(64)]
        fh.write(", \\\n".join(rows) + "\n")
    print("wrote", out)


if __name__ == "__main__":
    main()

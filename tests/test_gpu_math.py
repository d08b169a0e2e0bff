"""The kernels' own FP64 elementary functions against 50-digit references (mpmath)."""
import mpmath as mp
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
mp.mp.dps = 50
EPS = np.finfo(np.float64).eps


def run(afiv, op, x, y=None):
    xd = torch.tensor(x, dtype=torch.float64, device="cuda")
    yd = None if y is None else torch.tensor(y, dtype=torch.float64, device="cuda")
    out = afiv.debug_math(op, xd, yd)
    if op in ("sincos", "expm1", "cosm1", "expm1_tab"):
        return out[0].cpu().numpy(), out[1].cpu().numpy()
    return out.cpu().numpy()


def relerr(got, ref):
    return np.array([abs(float((mp.mpf(float(g)) - r) / r)) if r != 0 else abs(float(g)) for g, r in zip(got, ref)])


@pytest.mark.parametrize("which", ["exp", "exp_tab"])
def test_exp(afiv, which):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-700, 700, 3000), rng.uniform(-2, 2, 3000), [0.0, -745.0, 709.0, 1e-300]])
    if which == "exp_tab":                 # no clamp in the table form: finite arguments only
        got = run(afiv, which, x)
        ref = [mp.exp(mp.mpf(float(v))) for v in x]
        normal = np.array([float(r) > 1e-300 for r in ref])
        assert relerr(got[normal], [r for r, k in zip(ref, normal) if k]).max() <= 3 * EPS
        assert np.all(got[~normal] < 1e-300)
        edge = run(afiv, which, np.array([-1e6, -800.0, 800.0, np.nan]))
        assert edge[0] == 0 and edge[1] == 0 and np.isinf(edge[2]) and np.isnan(edge[3])
        return
    got = run(afiv, "exp", x)
    ref = [mp.exp(mp.mpf(float(v))) for v in x]
    normal = np.array([float(r) > 1e-300 for r in ref])        # subnormal results: absolute only
    assert relerr(got[normal], [r for r, k in zip(ref, normal) if k]).max() <= 3 * EPS
    assert np.all(got[~normal] < 1e-300)
    edge = run(afiv, "exp", np.array([-1e300, -800.0, 800.0, np.inf, -np.inf, np.nan]))
    assert edge[0] == 0 and edge[1] == 0 and np.isinf(edge[2]) and np.isinf(edge[3]) and edge[4] == 0
    assert np.isnan(edge[5])


def test_table_expm1_keeps_relative_accuracy_everywhere(afiv):
    """The table-driven exp(x) - 1 of the pricing kernel: 2^m T_j - 1 is exact where it cancels, so
    the relative accuracy holds at every argument, the bin edges around zero included."""
    rng = np.random.default_rng(17)
    edges = np.log(2) / 256 * np.array([1, -1, 0.999, -0.999, 1.001, -1.001, 3, -3, 255, -255, 257, -257])
    x = np.concatenate([rng.uniform(-30, 30, 2000), rng.uniform(-0.4, 0.4, 3000), rng.uniform(-0.02, 0.02, 3000),
                        10.0 ** rng.uniform(-300, -1, 2000) * rng.choice([-1.0, 1.0], 2000), edges])
    em1, e = run(afiv, "expm1_tab", x)
    assert relerr(em1, [mp.expm1(mp.mpf(float(v))) for v in x]).max() <= 4 * EPS
    assert relerr(e, [mp.exp(mp.mpf(float(v))) for v in x]).max() <= 3 * EPS
    em1, e = run(afiv, "expm1_tab", np.array([-800.0, -1e6]))
    assert np.all(em1 == -1.0) and np.all(e == 0.0)


def test_expm1_and_cosm1_keep_relative_accuracy_near_zero(afiv):
    """exp(x) - 1 and cos(x) - 1 as the pricing kernel forms 1 - exp(-d T): relative accuracy down to
    arguments of 1e-300, where subtracting 1 afterwards would leave nothing."""
    rng = np.random.default_rng(7)
    x = np.concatenate([rng.uniform(-30, 30, 2000), rng.uniform(-0.4, 0.4, 2000),
                        10.0 ** rng.uniform(-300, -1, 2000) * rng.choice([-1.0, 1.0], 2000), [0.34, -0.34, 0.35, -0.35]])
    em1, e = run(afiv, "expm1", x)
    assert relerr(em1, [mp.expm1(mp.mpf(float(v))) for v in x]).max() <= 4 * EPS
    assert relerr(e, [mp.exp(mp.mpf(float(v))) for v in x]).max() <= 3 * EPS
    y = np.concatenate([rng.uniform(-10, 10, 2000), rng.uniform(-0.06, 0.06, 2000),
                        10.0 ** rng.uniform(-150, -1, 2000) * rng.choice([-1.0, 1.0], 2000),
                        rng.uniform(-6000, 6000, 1000)])
    cm1, c = run(afiv, "cosm1", y)
    ref = [-2 * mp.sin(mp.mpf(float(v)) / 2) ** 2 for v in y]            # cos - 1 without cancellation
    # relative where the cosine is within 2^-8 of 1 (the table's first bin), absolute elsewhere
    near = np.abs(np.remainder(y + np.pi / 64, 2 * np.pi) - np.pi / 64) < np.pi / 64
    near &= np.abs(y) < 0.049
    assert relerr(cm1[near], [r for r, k in zip(ref, near) if k]).max() <= 4 * EPS
    assert np.max(np.abs(np.array([float(mp.mpf(float(g)) - r) for g, r in zip(cm1, ref)]))) <= 4e-16
    assert np.max(np.abs(np.array([float(mp.mpf(float(g)) - mp.cos(mp.mpf(float(v)))) for g, v in zip(c, y)]))) <= 3e-16


def test_sincos_absolute_accuracy(afiv):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-10, 10, 3000), rng.uniform(-6000, 6000, 3000),
                        rng.uniform(-1e6, 1e6, 2000), rng.uniform(-1e9, 1e9, 1000),
                        [0.0, np.pi / 4, np.pi / 2, -np.pi, 1e-300]])
    s, c = run(afiv, "sincos", x)
    es = np.array([abs(float(mp.mpf(float(a)) - mp.sin(mp.mpf(float(v))))) for a, v in zip(s, x)])
    ec = np.array([abs(float(mp.mpf(float(a)) - mp.cos(mp.mpf(float(v))))) for a, v in zip(c, x)])
    assert es.max() <= 3e-16 and ec.max() <= 3e-16


def test_log1p_relative_accuracy(afiv):
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(-0.9, 5, 3000), 10.0 ** rng.uniform(-20, 6, 2000),
                        -(10.0 ** rng.uniform(-20, -0.2, 2000)), [0.0, 1e-300, -0.29, 0.41, 1.0, 3.0]])
    got = run(afiv, "log1p", x)
    ref = [mp.log1p(mp.mpf(float(v))) for v in x]
    assert relerr(got, ref).max() <= 3 * EPS


def test_atan2_all_quadrants(afiv):
    rng = np.random.default_rng(3)
    n = 6000
    y = rng.standard_normal(n) * 10.0 ** rng.uniform(-12, 3, n)
    x = rng.standard_normal(n) * 10.0 ** rng.uniform(-12, 3, n)
    y = np.concatenate([y, [0.0, 1.0, -1.0, 1e-300, 0.4142135623730950, 1.0]])
    x = np.concatenate([x, [-1.0, 0.0, 0.0, 1.0, 1.0, 1.0]])
    got = run(afiv, "atan2", y, x)
    ref = [mp.atan2(mp.mpf(float(a)), mp.mpf(float(b))) for a, b in zip(y, x)]
    assert relerr(got, ref).max() <= 3 * EPS


def test_table_driven_functions_at_their_bin_edges(afiv):
    """sincos, log1p and atan2 reduce against tables (pi/32 steps; 128 bins of the mantissa laid out
    around 1.0; 1/64 steps of the quotient): the points where the bin index changes, where the
    rounding of the index is a tie, and the neighbourhood of the bins that carry exact entries."""
    import struct
    ulps = np.array([-2.0, -1.0, 0.0, 1.0, 2.0])
    # log1p: u = 1 + f on both sides of every bin edge, over a few binades, and around u = 1
    edges = [struct.unpack("<d", struct.pack("<Q", (0x3fe5f000 + i * 0x2000) << 32))[0] for i in range(129)]
    u = np.array([np.nextafter(e, np.inf if d > 0 else -np.inf) if d else e for e in edges for d in (-1, 0, 1)])
    f = np.concatenate([(u * 2.0 ** k) - 1.0 for k in (-2, -1, 0, 1, 5)] +
                       [np.array([s * 2.0 ** -e for e in (8, 9, 10, 30, 52, 60, 200)
                                  for s in (1.0, -1.0, 1 + 2.0 ** -30, -(1 - 2.0 ** -30))])])
    f = f[f > -1.0]
    got = run(afiv, "log1p", f)
    ref = [mp.log1p(mp.mpf(float(v))) for v in f]
    nz = np.array([r != 0 for r in ref])
    assert relerr(got[nz], [r for r, k in zip(ref, nz) if k]).max() <= 3 * EPS
    assert np.all(got[~nz] == 0.0)
    # atan2: quotients at j/64 and (j + 1/2)/64, a few ulp either side, in every octant
    tq = np.concatenate([(j / 64.0) * (1 + ulps * EPS) for j in range(1, 65)] +
                        [((j + 0.5) / 64.0) * (1 + ulps * EPS) for j in range(64)])
    ys, xs = [], []
    for sy, sx, swap in ((1, 1, 0), (1, -1, 0), (-1, 1, 0), (-1, -1, 0), (1, 1, 1), (-1, -1, 1)):
        a, b = (tq, np.ones_like(tq)) if not swap else (np.ones_like(tq), tq)
        ys.append(sy * a * 3.7); xs.append(sx * b * 3.7)
    y, x = np.concatenate(ys), np.concatenate(xs)
    got = run(afiv, "atan2", y, x)
    ref = [mp.atan2(mp.mpf(float(a)), mp.mpf(float(b))) for a, b in zip(y, x)]
    assert relerr(got, ref).max() <= 3 * EPS
    # sincos: multiples and half-multiples of pi/32, near and far
    m = np.concatenate([np.arange(-200, 201), [10 ** 4, 10 ** 6 + 17, -(10 ** 7) - 3]]).astype(np.float64)
    xx = np.concatenate([(m * np.pi / 32) * (1 + d * EPS) for d in ulps] +
                        [((m + 0.5) * np.pi / 32) * (1 + d * EPS) for d in ulps])
    sn, cs = run(afiv, "sincos", xx)
    es = np.array([abs(float(mp.mpf(float(a)) - mp.sin(mp.mpf(float(v))))) for a, v in zip(sn, xx)])
    ec = np.array([abs(float(mp.mpf(float(a)) - mp.cos(mp.mpf(float(v))))) for a, v in zip(cs, xx)])
    assert es.max() <= 3e-16 and ec.max() <= 3e-16


def test_rcp(afiv):
    rng = np.random.default_rng(4)
    x = rng.standard_normal(4000) * 10.0 ** rng.uniform(-100, 100, 4000)
    got = run(afiv, "rcp", x)
    assert np.max(np.abs(got * x - 1.0)) <= 3 * EPS


def test_ndtr_against_reference(afiv):
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-8, 8, 6000), rng.uniform(-38, -8, 1000), [0.0, -0.0, 40.0, -40.0, np.inf, -np.inf]])
    got = run(afiv, "ndtr", x)
    ref = [mp.ncdf(mp.mpf(float(v))) for v in x]
    refd = np.array([float(r) for r in ref])
    # absolute: an ulp of 1.0; relative in the lower tail: a few ulp plus the rounding of the
    # argument's own scaling by 1/sqrt(2), which exp(-x^2/2) amplifies by x^2 (1.2e-13 at x = -33)
    assert np.max(np.abs(got - refd)) <= 2.5e-16
    tail = (x < 0) & (refd > 1e-300)
    rel = relerr(got[tail], [r for r, t in zip(ref, tail) if t])
    assert np.all(rel <= 4 * EPS * (1 + x[tail] ** 2))
    assert np.isnan(run(afiv, "ndtr", np.array([np.nan]))[0])

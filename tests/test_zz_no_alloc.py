"""afiv_surface_static (the entry point that allocates nothing: price ratios pass through a fixed
array inside the library) against afiv_surface, through the C ABI: two calls in a row, the second
on a smaller batch, must each equal afiv_surface bit for bit; a batch beyond the array passes through
it in pieces; one beyond the stated number of pieces is rejected before anything is launched."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SPOT, R, Q = 21000.0, 0.067, 0.0
SCRATCH_DOUBLES = 16 * 1024 * 1024


def _params(n, seed):
    from test_gpu_parity import sample_params
    return torch.as_tensor(sample_params(n, seed)).cuda()


def _static(afiv, grid, p, variant):
    n = p.shape[0]
    iv = torch.full((n, grid.nT, grid.nK), -7.0, dtype=torch.float64, device="cuda")
    fl = torch.full((n, 4), -7, dtype=torch.int32, device="cuda")
    rc = afiv._lib.lib().afiv_surface_static(grid._h, ctypes.c_void_p(p.data_ptr()), n, SPOT, variant,
                                            ctypes.c_void_p(iv.data_ptr()), ctypes.c_void_p(fl.data_ptr()),
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    return rc, iv, fl


def _same(a, b):
    return torch.equal(torch.nan_to_num(a, nan=-1.0), torch.nan_to_num(b, nan=-1.0))


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_static_scratch_entry_point_equals_afiv_surface(afiv, variant):
    logm = np.linspace(-0.2, 0.2, 50)
    tau = np.round(np.linspace(30, 730, 20)) / 365.0
    grid = afiv.Grid(logm, tau, R, Q)
    for n, seed in ((3000, 5), (37, 6)):            # the second call reuses the array with a smaller batch
        p = _params(n, seed)
        rc, iv, fl = _static(afiv, grid, p, variant)
        assert rc == 0
        iv_ref, fl_ref, _ = grid.surface(p, SPOT, variant)
        assert _same(iv, iv_ref) and torch.equal(fl, fl_ref)
        assert (fl[:, 3] & 4).any()


def test_static_scratch_entry_point_limits(afiv):
    grid = afiv.Grid(np.linspace(-0.2, 0.2, 21), [0.25, 0.5, 1.0, 2.0], R, Q)
    per = 4 * 21
    most = SCRATCH_DOUBLES // per                    # what the array holds at once
    piece = most // 1792 * 1792                      # the size of a piece: a multiple of 7 x 256 surfaces
    # a batch of a little over two array-fulls goes through in three pieces: same results
    for n in (most, most + 1, 2 * piece + 5):
        p = _params(n, 9)
        rc, iv, fl = _static(afiv, grid, p, 0)
        assert rc == 0
        iv_ref, fl_ref, _ = grid.surface(p, SPOT, 0)
        assert _same(iv, iv_ref) and torch.equal(fl, fl_ref), n
    # beyond 1024 pieces: rejected before anything is launched (the buffers are never touched, so
    # small ones will do)
    p = _params(64, 9)
    iv = torch.full((64, 4, 21), -7.0, dtype=torch.float64, device="cuda")
    fl = torch.full((64, 4), -7, dtype=torch.int32, device="cuda")
    rc = afiv._lib.lib().afiv_surface_static(grid._h, ctypes.c_void_p(p.data_ptr()), 1024 * piece + 1, SPOT, 0,
                                            ctypes.c_void_p(iv.data_ptr()), ctypes.c_void_p(fl.data_ptr()),
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert rc == -6 and "scratch" in afiv._lib.lib().afiv_error_string(rc).decode()
    assert float(iv.max()) == -7.0 and int(fl.max()) == -7
    rc, _, _ = _static(afiv, grid, p[:0].contiguous(), 0)
    assert rc == 0


def test_static_scratch_entry_point_at_the_benchmark_batch(afiv):
    """131 072 surfaces of 20 x 50 (1 GB of ratios) through the 128 MiB array: 9 pieces, the vols and
    flag words of the two-call path, bit for bit."""
    logm = np.linspace(-0.2, 0.2, 50)
    tau = np.round(np.linspace(30, 730, 20)) / 365.0
    grid = afiv.Grid(logm, tau, R, Q)
    p = _params(131072, 21)
    rc, iv, fl = _static(afiv, grid, p, 0)
    assert rc == 0
    ratio = grid.price(p)
    iv_ref, fl_ref = grid.implied_vol_checked(ratio, SPOT, "A")
    assert _same(iv, iv_ref) and torch.equal(fl, fl_ref)


@pytest.mark.parametrize("nT,nK,variant", [(8, 21, "A"), (64, 64, "A"), (64, 64, "B"), (20, 50, "A_BRENT")])
def test_static_scratch_entry_point_under_graph_capture(afiv, nT, nK, variant):
    """What the entry point is for: a whole price -> implied vol -> checks pass captured into a HIP
    graph (capture forbids allocation) and replayed on new parameters in the same buffers.  The
    largest lattice makes the kernels ask for more than the default dynamic LDS, which is a function
    attribute set at launch time -- legal under capture, unlike an allocation."""
    if (nT, nK) == (8, 21):
        logm = np.linspace(-0.2, 0.2, 21)
        tau = np.array([0.08333333, 0.16666667, 0.25, 0.5, 0.75, 1.0, 1.5, 2.0])
    else:
        logm = np.linspace(-0.3, 0.25, nK)
        tau = np.round(np.linspace(20, 900, nT)) / 365.0
    grid = afiv.Grid(logm, tau, R, Q)
    n = 500
    p = _params(n, 11).clone()
    iv = torch.empty((n, nT, nK), dtype=torch.float64, device="cuda")
    fl = torch.empty((n, 4), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):            # the first call on this grid is the captured one
        grid.surface_static(p, SPOT, variant, iv_out=iv, flags_out=fl)
    for seed in (12, 13):
        p.copy_(_params(n, seed))
        graph.replay()
        torch.cuda.synchronize()
        iv_ref, fl_ref, _ = grid.surface(p, SPOT, variant)
        assert _same(iv, iv_ref) and torch.equal(fl, fl_ref)

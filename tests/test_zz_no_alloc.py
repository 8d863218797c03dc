"""tad_metadata_noise_static (the entry point that allocates nothing: column statistics in a device array of the
library) against tad_metadata_noise (caller-provided workspace), through the C ABI: the same results (to about one
unit in the last place: the column statistics are sums of float64 atomics, whose order differs from run to run) on two
calls in a row, the second with a smaller input, so stale contents of the shared array would show."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(lib, name, cond, cols, noise, level, stats=None):
    rows, F = cond.shape
    out = torch.full_like(cond, float("nan"))
    carr = (C.c_int32 * len(cols))(*cols)
    stream = torch.cuda.current_stream().cuda_stream
    if stats is None:
        rc = getattr(lib, name)(rows, F, len(cols), carr, cond.data_ptr(), cond.stride(0), noise.data_ptr(), float(level),
                                out.data_ptr(), out.stride(0), stream)
    else:
        rc = getattr(lib, name)(rows, F, len(cols), carr, cond.data_ptr(), cond.stride(0), noise.data_ptr(), float(level),
                                stats.data_ptr(), out.data_ptr(), out.stride(0), stream)
    assert rc == 0, lib.tad_last_error().decode()
    torch.cuda.synchronize()
    return out


def test_metadata_noise_static_matches_workspace_form(lib):
    from oracle import tvmcg as otv
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    rng = np.random.default_rng(77)
    # first call: 64 selected columns (the maximum) of a large table; second call: fewer rows and columns
    for rows, F, ncol, level in [(24 * 4096, 70, 64, 0.4), (24 * 37, 28, 5, 0.25), (1, 6, 2, 0.1)]:
        cond_h = (rng.standard_normal((rows, F)) * rng.uniform(0.1, 5.0, F)).astype(np.float32)
        cols = [int(c) for c in rng.permutation(F)[:ncol]]
        noise_h = rng.standard_normal((rows, ncol)).astype(np.float32)
        cond, noise = torch.from_numpy(cond_h).to(dev), torch.from_numpy(noise_h).to(dev)
        stats = torch.empty(2 * ncol, dtype=torch.float64, device=dev)
        ref = _run(lib, "tad_metadata_noise", cond, cols, noise, level, stats)
        got = _run(lib, "tad_metadata_noise_static", cond, cols, noise, level)
        # float64 atomics of the column statistics arrive in any order: the rounded float32 standard deviation
        # may differ in its last bit between any two launches, so compare to 1 ulp-scale and to the oracle
        assert torch.equal(torch.isnan(got), torch.isnan(ref))
        np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=2e-7, atol=1e-7, equal_nan=True)
        if rows > 1:
            want = otv.add_continuous_metadata_noise_subset(cond_h[None], cols, level, noise_h[None])[0]
            np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-6, atol=1e-6)
        untouched = [c for c in range(F) if c not in cols]
        assert torch.equal(got[:, untouched], cond[:, untouched])


def test_metadata_noise_static_rejects_oversize_and_empty(lib):
    dev = torch.device("cuda:0")
    cond = torch.zeros(8, 80, device=dev)
    noise = torch.zeros(8, 65, device=dev)
    out = torch.empty_like(cond)
    cols = (C.c_int32 * 65)(*range(65))
    rc = lib.tad_metadata_noise_static(8, 80, 65, cols, cond.data_ptr(), 80, noise.data_ptr(), 0.5, out.data_ptr(), 80,
                                       torch.cuda.current_stream().cuda_stream)
    assert rc != 0 and "at most 64" in lib.tad_last_error().decode()
    # no rows: nothing to do, no launch
    rc = lib.tad_metadata_noise_static(0, 80, 3, cols, cond.data_ptr(), 80, noise.data_ptr(), 0.5, out.data_ptr(), 80,
                                       torch.cuda.current_stream().cuda_stream)
    assert rc == 0


def test_metadata_noise_static_replayed_from_a_graph(lib):
    """The callers this entry point is for include code under graph capture: captured once and replayed between other work
    of the process (fills, small copies), every replay gives the result of a direct call (the statistics are cleared by a
    kernel of the library: a memset node was seen to write other values than the captured zero on replay)."""
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    rng = np.random.default_rng(3)
    rows, F, cols, level = 24 * 200, 28, [3, 9, 27, 0, 14], 0.3
    cond = torch.from_numpy((rng.standard_normal((rows, F)) * rng.uniform(0.1, 5.0, F)).astype(np.float32)).to(dev)
    noise = torch.from_numpy(rng.standard_normal((rows, len(cols))).astype(np.float32)).to(dev)
    ref = _run(lib, "tad_metadata_noise_static", cond, cols, noise, level)
    out = torch.empty_like(cond)
    carr = (C.c_int32 * len(cols))(*cols)
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        with torch.cuda.graph(g, stream=cap):
            rc = lib.tad_metadata_noise_static(rows, F, len(cols), carr, cond.data_ptr(), cond.stride(0), noise.data_ptr(), float(level),
                                               out.data_ptr(), out.stride(0), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.tad_last_error().decode()
    for rep in range(8):
        assert int((cond > 0.2 * rep).nonzero().shape[0]) > 0
        out.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), rtol=2e-7, atol=1e-7)

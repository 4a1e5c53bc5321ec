"""shivae_sqnorm_noalloc (the entry point that allocates nothing: its block partials live in a file-scope device
array) against shivae_sqnorm, through the C ABI: two calls in a row, the second on a smaller input."""
import numpy as np
import pytest
import torch

from shivae_amd import _native as N

pytestmark = pytest.mark.gpu


def _both(g):
    L = N.lib()
    n = g.numel()
    scratch = torch.zeros(1024, device="cuda")
    ref, out = torch.full((1,), -1.0, device="cuda"), torch.full((1,), -2.0, device="cuda")
    N.check(L.shivae_sqnorm(N.ptr(g), n, N.ptr(scratch), N.ptr(ref), N.stream()))
    N.check(L.shivae_sqnorm_noalloc(N.ptr(g), n, N.ptr(out), N.stream()))
    torch.cuda.synchronize()
    return float(ref), float(out)


def test_sqnorm_noalloc_matches_sqnorm_twice_in_a_row():
    gen = torch.Generator(device="cuda").manual_seed(7)
    big = torch.randn(3_000_017, device="cuda", generator=gen) * 3
    small = torch.randn(1_237, device="cuda", generator=gen)
    for g in (big, small):  # the second call re-uses the array the first one filled (1024 partials -> 5)
        ref, out = _both(g)
        assert out == ref, (g.numel(), out, ref)  # same kernels, same partials: bit-identical
        np.testing.assert_allclose(out, float((g.double() ** 2).sum()), rtol=1e-5)


def test_sqnorm_noalloc_rejects_oversize():
    L = N.lib()
    g = torch.zeros(4, device="cuda")
    out = torch.zeros(1, device="cuda")
    rc = L.shivae_sqnorm_noalloc(N.ptr(g), (1 << 32) + 1, N.ptr(out), N.stream())
    assert rc != 0 and b"exceed" in L.shivae_last_error()

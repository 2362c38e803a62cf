"""csrc/nvbx_arith.h: the device's shortened division / square-root sequences give the IEEE results (the CPU oracle uses the plain
operators, so bit parity of every projection rests on this)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(hip_lib, a, b):
    import torch
    ta = torch.from_numpy(a).cuda(); tb = torch.from_numpy(b).cuda()
    q = torch.empty_like(ta); r = torch.empty_like(ta)
    rc = hip_lib.nvbx_selftest_arith(ta.data_ptr(), tb.data_ptr(), q.data_ptr(), r.data_ptr(), a.size)
    assert rc == 0
    return q.cpu().numpy(), r.cpu().numpy()


def test_division_and_square_root_are_the_ieee_results(hip_lib):
    rng = np.random.default_rng(7)
    n = 1 << 24
    # magnitudes over the documented range 2^-60 .. 2^60, random signs and mantissas; plus the values the kernels really see
    mag = lambda: np.exp2(rng.uniform(-60.0, 60.0, n)).astype(np.float32) * rng.choice(np.float32([-1.0, 1.0]), n)
    for a, b in ((mag(), mag()),
                 (rng.uniform(-200.0, 200.0, n).astype(np.float32), rng.uniform(1e-3, 200.0, n).astype(np.float32)),
                 (rng.uniform(-1.0, 1.0, n).astype(np.float32), rng.uniform(0.5, 2.0, n).astype(np.float32))):
        q, r = _run(hip_lib, a, b)
        with np.errstate(all="ignore"):
            assert np.array_equal(q.view(np.uint32), (a / b).view(np.uint32))
            assert np.array_equal(r.view(np.uint32), np.sqrt(np.abs(a)).view(np.uint32))


def test_special_operands(hip_lib):
    a = np.float32([0.0, -0.0, 1.0, -1.0, np.inf, 0.0, 3.0, 1.0, 4.0, 2.0 ** -96, 2.0 ** 100, 1e-30, 1.5])
    b = np.float32([2.0, 2.0, np.inf, 0.0, 2.0, 0.0, -0.0, 3.0, 2.0, 1.0, 1.0, 1.0, np.nan])
    q, r = _run(hip_lib, a, b)
    with np.errstate(all="ignore"):
        want = a / b
        assert np.array_equal(np.isnan(q), np.isnan(want))
        ok = ~np.isnan(want)
        assert np.array_equal(q[ok].view(np.uint32), want[ok].view(np.uint32))          # signed zeros and infinities included
        fin = np.isfinite(a)
        assert np.array_equal(r[fin].view(np.uint32), np.sqrt(np.abs(a[fin])).view(np.uint32))

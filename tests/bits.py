"""Helper of the parity tests: float32 arrays equal in their BITS (NaN-safe, -0.0 != +0.0), with a useful message."""
import numpy as np


def assert_same_bits(got, expect, err_msg=''):
    got, expect = np.asarray(got), np.asarray(expect)
    assert got.dtype == expect.dtype == np.float32 and got.shape == expect.shape, (err_msg, got.dtype, got.shape, expect.dtype, expect.shape)
    bad = got.view(np.uint32) != expect.view(np.uint32)
    assert not bad.any(), '%s: %d of %d values differ in their bits (max |diff| %g)' % (
        err_msg, int(bad.sum()), bad.size, float(np.abs(got.astype(np.float64) - expect).max()))

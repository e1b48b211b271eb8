"""Race / hazard screen of the persistent body convolution (the kernel with hand-ordered pipelines): many
launches on fresh random data at 1..8 items per workgroup, every output element compared with the simple
one-tile-per-workgroup structure, plus bitwise run-to-run determinism.  (guide: "screen a sync-structure edit for
races over many runs at several sizes")."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_body_conv_matches_reference_structure_over_many_runs():
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'stress_body_conv.py')], capture_output=True,
                       text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert 'total mismatching elements = 0' in p.stdout


def test_bf16_body_conv_matches_reference_structure_over_many_runs():
    """Same screen for the bf16 kernel (F = 256 and 128) against the fp32 reference structure on bf16-exact operands
    (summation-order tolerance), plus bit-exact consistency of the residual planes."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'stress_body_conv_bf16.py')], capture_output=True,
                       text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert 'total mismatching elements = 0' in p.stdout


def test_bf16x3_body_conv_matches_reference_structure_over_many_runs():
    """Same screen for the bf16x3 kernel against the fp32 reference structure on the fp32 operands, plus bit-exact consistency
    of the stream's three planes."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'stress_body_conv_bf16x3.py')], capture_output=True,
                       text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert 'total mismatching elements = 0' in p.stdout


def test_forward_is_bitwise_reproducible_across_launches():
    from dsen2_amd import weights as W
    from dsen2_amd.DSen2Net import s2model
    for prec in ('fp32', 'bf16', 'bf16x3'):
        m = s2model(((4, None, None), (6, None, None)), num_layers=6, feature_size=128, precision=prec)
        m.set_weights_flat(W.random_he_uniform(10, 6, 6, 128, seed=5, bias_scale=0.05))
        xs = [torch.rand((300, 4, 32, 32), device='cuda') * 5, torch.rand((300, 6, 32, 32), device='cuda') * 5]
        ref = m.forward_device(xs).clone()
        for _ in range(8):
            assert torch.equal(m.forward_device(xs), ref), prec


@pytest.mark.parametrize('feat,d,n,prec', [(256, 8, 256, 'bf16'), (128, 4, 256, 'bf16'), (128, 4, 512, 'bf16'), (128, 4, 511, 'bf16'),
                                           (128, 3, 512, 'bf16x3'), (128, 3, 511, 'bf16x3'), (256, 3, 256, 'bf16x3')])
def test_chain_kernel_over_many_launches_on_fresh_data(feat, d, n, prec):
    """Race screen of the chain kernel's layer boundaries (seamless with one patch per workgroup at F = 256 and two at
    F = 128, drained at F = 128 with one): the chain launch depends on hand-placed waits for its OWN stores between
    layers, so a too-weak one shows up as a rare stale halo.  60 forwards on fresh random inputs each, every one
    compared bit for bit with the same patches run layer by layer (sub-batches of 5: the per-layer kernels)."""
    from dsen2_amd import weights as W
    from dsen2_amd.DSen2Net import s2model
    m = s2model(((4, None, None), (6, None, None)), num_layers=d, feature_size=feat, precision=prec)
    m.set_weights_flat(W.random_he_uniform(10, 6, d, feat, seed=feat + d, bias_scale=0.05))
    assert m.body_launches(n, 32, 32) == 1 and m.body_launches(5, 32, 32) == 2 * d
    gen = torch.Generator(device='cuda').manual_seed(n + feat)
    bad = 0
    for rep in range(60 if prec == 'bf16' else 30):
        xs = [torch.rand((n, c, 32, 32), device='cuda', generator=gen) * 5 for c in (4, 6)]
        y = m.forward_device(xs)
        for first in (0, (37 * rep) % (n - 5), n - 5):
            sub = m.forward_device([t[first:first + 5].contiguous() for t in xs])
            bad += int((sub != y[first:first + 5]).sum())
    assert bad == 0


def test_output_conv_fuzz_against_the_oracle_and_the_reference_structure():
    """conv3x3_out_mfma.hip over random shapes (one or several 32-pixel blocks per row, 1-4 rows per wave, strips of rows, Cout
    1..6, F 128 / 256): every element against the float64 oracle; then many launches on fresh data at bench-like sizes against
    the independent one-tile-per-workgroup kernel (dsen2_conv3x3_nhwc_ref), and run-to-run bit identity."""
    import numpy as np
    from dsen2_amd.DSen2Net import conv3x3_nhwc
    from oracle import c_oracle
    rng = np.random.default_rng(2024)
    for it in range(24):
        feat = (128, 256)[it % 2]
        cout = int(rng.integers(1, 7))
        n, h, w = int(rng.integers(1, 5)), int(rng.integers(1, 75)), int(rng.integers(1, 140))
        x = rng.standard_normal((n, feat, h, w)).astype(np.float32)
        skip = rng.standard_normal((n, cout, h, w)).astype(np.float32)
        k = (rng.standard_normal((3, 3, feat, cout)) * np.sqrt(2.0 / (9 * feat))).astype(np.float32)
        b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
        xd = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 3, 1))).cuda()
        y = conv3x3_nhwc(xd, k, b, epilogue=2, aux=torch.from_numpy(skip).cuda()).cpu().numpy()
        ref = c_oracle.conv3x3(x, k, b) + skip
        assert np.abs(y - ref).max() < 2e-5, (feat, cout, n, h, w, float(np.abs(y - ref).max()))
    g = torch.Generator(device='cuda'); g.manual_seed(7)
    worst = 0.0
    for it in range(30):
        feat, cout = ((128, 6), (128, 2), (256, 6))[it % 3]
        n, h, w = ((512, 32, 32), (40, 128, 128), (300, 32, 32), (7, 192, 192), (64, 64, 48))[it % 5]
        if feat == 256:
            n = max(1, n // 2)
        xd = torch.randn((n, h, w, feat), device='cuda', generator=g)
        sd = torch.randn((n, cout, h, w), device='cuda', generator=g)
        k = (rng.standard_normal((3, 3, feat, cout)) * np.sqrt(2.0 / (9 * feat))).astype(np.float32)
        b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
        a = conv3x3_nhwc(xd, k, b, epilogue=2, aux=sd)
        a2 = conv3x3_nhwc(xd, k, b, epilogue=2, aux=sd)
        r = conv3x3_nhwc(xd, k, b, epilogue=2, aux=sd, ref=True)
        assert torch.equal(a, a2), (it, 'not deterministic')
        worst = max(worst, float((a - r).abs().max()))
        assert worst < 4e-5, (it, feat, cout, n, h, w, worst)
        del xd, sd, a, a2, r

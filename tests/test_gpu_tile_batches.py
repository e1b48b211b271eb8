"""Which batches run the residual blocks as ONE chain launch, and the batch size predict() / the tile path choose
(S2Model.preferred_batch).  The chain kernel is taken for patches of at most 8 tiles (64 x 64) — measured 1-3 % ahead of the
per-layer launches there and 1-5 % behind them on the 128^2 / 192^2 patches of a real tile
(profiles/r04_k_chain_vs_layerwise.txt) — so a tile's patches always go layer by layer, whatever the batch, and small patches
are cut into multiples of the CU count."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('precision', ['bf16', 'bf16x3'])
def test_chain_for_small_patches_only_and_the_batch_follows(precision):
    from dsen2_amd import weights as W
    from dsen2_amd.DSen2Net import s2model
    m = s2model(((4, None, None), (6, None, None)), num_layers=6, feature_size=128, precision=precision)
    m.set_weights_flat(W.random_he_uniform(10, 6, 6, 128, seed=4, bias_scale=0.05))
    cus = int(torch.cuda.get_device_properties(0).multi_processor_count)
    for p in (32, 64):                                                   # chained: the preferred batch is whole patches per workgroup
        pb = m.preferred_batch(p, p)
        assert pb % cus == 0 and cus <= pb <= m.batch_limit(p, p), (p, pb, m.batch_limit(p, p))
        assert m.body_launches(pb, p, p) == 1
    for p in (96, 128, 192):                                             # a real tile's patches: layer by layer for every batch
        assert m.preferred_batch(p, p) == m.batch_limit(p, p)
        for n in (cus, 2 * cus, 300):
            assert m.body_launches(n, p, p) == 12, (p, n)
    # predict() (host arrays in, batches chosen inside) does not depend on how it cuts: 2 x cus + 7 patches of 32^2 in one
    # go (one chain launch + a short per-layer batch) against explicit batches of 100
    rng = np.random.default_rng(6)
    n = 2 * cus + 7
    xs = [rng.random((n, c, 32, 32), dtype=np.float32) * np.float32(5) for c in (4, 6)]
    m.max_workspace_bytes = m.workspace_bytes(2 * cus + 3, 32, 32)       # limit 2 cus + 3 -> preferred 2 cus
    assert m.preferred_batch(32, 32) == 2 * cus
    a = m.predict(xs)
    b = m.predict(xs, batch_size=100)
    assert np.array_equal(a, b) and np.isfinite(a).all()


def test_fp32_keeps_the_memory_bound_batch():
    from dsen2_amd.DSen2Net import s2model
    m = s2model(((4, None, None), (6, None, None)), num_layers=6, feature_size=128)
    assert m.preferred_batch(128, 128) == m.batch_limit(128, 128)
    assert m.preferred_batch(32, 32) == m.batch_limit(32, 32)
    b = s2model(((4, None, None), (6, None, None)), num_layers=6, feature_size=128, precision='bf16')
    b.max_workspace_bytes = 64 << 20                                    # fewer patches than CUs fit: nothing to round to
    assert b.preferred_batch(32, 32) == b.batch_limit(32, 32) >= 1

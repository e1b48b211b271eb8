"""Whole-network outputs of one source tree as an .npz, for bit-for-bit comparison of two trees / libraries in one gpurun call.
    python tools/dump_forward_bits.py <repo root> <out.npz>"""
import os, sys, numpy as np, torch
root = sys.argv[1]; sys.path.insert(0, root)
from dsen2_amd.DSen2Net import s2model
from dsen2_amd import weights as W
res = {}
for bands, d, f in [((4, 6), 1, 128), ((4, 6, 2), 1, 128), ((4, 6), 1, 256)]:
    flat = W.random_he_uniform(sum(bands), bands[-1], d, f, seed=3, bias_scale=0.05)
    rng = np.random.default_rng(1)
    xs = [torch.from_numpy((rng.random((5, c, 37, 50), dtype=np.float32) * 5)).cuda() for c in bands]
    m = s2model(tuple((b, None, None) for b in bands), num_layers=d, feature_size=f)
    m.set_weights_flat(flat)
    res['%s_%d' % (len(bands), f)] = m.forward_device(xs).cpu().numpy()
np.savez(sys.argv[2], **res)

#!/usr/bin/env python3
"""Stage breakdown of DSen2_20 on a full synthetic tile (host ndarray in -> host ndarray out)."""
import contextlib, io, json, os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsen2_amd import supres, weights, patches as P, dist as D
n = 10980
rng = np.random.default_rng(0)
d10 = rng.integers(35, 13110, size=(n, n, 4), dtype=np.uint16)
d20 = rng.integers(35, 13110, size=(n // 2, n // 2, 6), dtype=np.uint16)
tmp = tempfile.mkdtemp(); np.save(os.path.join(tmp, 's2_032_lr_1e-04.npy'), weights.random_he_uniform(10, 6, 6, 128, seed=11))
supres.MDL_PATH = os.path.join(tmp, '')
with contextlib.redirect_stdout(io.StringIO()):
    supres.DSen2_20(d10[:240, :240], d20[:120, :120])
def T():
    torch.cuda.synchronize(); return time.perf_counter()
out = {}
t0 = T(); a = np.ascontiguousarray(d10, dtype=np.float32); b = np.ascontiguousarray(d20, dtype=np.float32); t1 = T(); out['host_to_f32_s'] = t1 - t0
dev = P.default_device()
x10 = torch.from_numpy(a).to(dev); x20 = torch.from_numpy(b).to(dev); t2 = T(); out['h2d_s'] = t2 - t1
x10u = torch.from_numpy(d10.view(np.int16)).to(dev); t2b = T(); out['h2d_uint16_only_10m_s'] = t2b - t2
org, n_alloc = P.tile_origins(x20.shape, 64, 4); used = org.shape[0]
with contextlib.redirect_stdout(io.StringIO()):
    model = supres._get_model(((4, None, None), (6, None, None)), False, False)
bs = model.preferred_batch(128, 128)
pred = torch.empty((used, 6, 128, 128), device=dev)
tg = tu = tf = 0.0
for i0 in range(0, used, bs):
    nb = min(bs, used - i0)
    s0 = T(); p10 = P.gather_patches_device(x10, org, 2, 8, 128, n_alloc, divisor=2000, first=i0, count=nb)
    lr = P.gather_patches_device(x20, org, 1, 4, 64, n_alloc, first=i0, count=nb); s1 = T()
    p20 = P.interp_patches_device(lr, (128, 128), post_divisor=2000); s2 = T()
    model.forward_device([p10, p20], out=pred[i0:i0 + nb]); s3 = T()
    tg += s1 - s0; tu += s2 - s1; tf += s3 - s2
out.update(gather_s=tg, upsample_s=tu, forward_s=tf, batch=bs, patches=used)
s0 = T(); img = P.recompose_device(pred, 8, (n, n), scale=2000); s1 = T(); out['recompose_s'] = s1 - s0
y = img.cpu().numpy(); s2 = T(); out['d2h_s'] = s2 - s1
print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in out.items()}))

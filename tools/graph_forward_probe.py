"""The whole forward replayed as ONE hipGraph (torch.cuda.CUDAGraph around dsen2_model_forward) against plain
launches: same bits, no gain (profiles/archive/r02_ablation.md §3) — the launches are not host-bound.
    python tools/graph_forward_probe.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dsen2_amd import weights as W
from dsen2_amd.DSen2Net import s2model
import os
cfgs = {'vdsen2_bf16': (32, 256, 'bf16', 256), 'dsen2_fp32': (6, 128, 'fp32', 512)}
for name, (D, F, prec, B) in cfgs.items():
    m = s2model(((4, None, None), (6, None, None)), num_layers=D, feature_size=F, precision=prec)
    m.set_weights_flat(W.random_he_uniform(10, 6, D, F, seed=1))
    xs = [torch.randn((B, 4, 32, 32), device='cuda'), torch.randn((B, 6, 32, 32), device='cuda')]
    out = torch.empty((B, 6, 32, 32), device='cuda')
    for _ in range(3): m.forward_device(xs, out=out)
    torch.cuda.synchronize()
    def timeit(fn, n=20):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        fn(); torch.cuda.synchronize()
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    t_plain = timeit(lambda: m.forward_device(xs, out=out))
    ref = out.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    try:
        with torch.cuda.stream(s):
            m.forward_device(xs, out=out)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            m.forward_device(xs, out=out)
        out.zero_()
        g.replay(); torch.cuda.synchronize()
        same = bool(torch.equal(out, ref))
        t_graph = timeit(lambda: g.replay())
        print(name, 'plain %.4f ms  graph %.4f ms  same bits %s' % (t_plain, t_graph, same), flush=True)
    except Exception as ex:
        print(name, 'plain %.4f ms  graph capture failed: %r' % (t_plain, ex), flush=True)

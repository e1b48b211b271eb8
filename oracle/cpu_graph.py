"""TEST INFRASTRUCTURE ONLY — the reference graph on the CPU in float32 via torch (oneDNN) conv2d.

This is the `cpu_baseline` ("kind": "port") of bench.py: the reference's own CPU path is keras/TensorFlow,
which is not installed anywhere in the build image and may not travel to the GPU box, so the same graph
(utils/DSen2Net.py:9-43) is restated with torch.nn.functional.conv2d, NCHW float32, all host cores.
Never imported by the product package.
"""
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import dsen2_oracle as do


def build(flat, cin, cout, num_layers, feature_size):
    layers = []
    for k, b in do.split_weights(np.asarray(flat), cin, cout, num_layers, feature_size):
        w = torch.from_numpy(np.ascontiguousarray(k.transpose(3, 2, 0, 1)))       # HWIO -> OIHW
        layers.append((w, torch.from_numpy(np.ascontiguousarray(b))))
    return layers


def forward(layers, xs, num_layers):
    x = torch.cat(xs, dim=1)
    x = F.relu(F.conv2d(x, layers[0][0], layers[0][1], padding=1))
    for i in range(num_layers):
        t = F.relu(F.conv2d(x, layers[1 + 2 * i][0], layers[1 + 2 * i][1], padding=1))
        x = x + F.conv2d(t, layers[2 + 2 * i][0], layers[2 + 2 * i][1], padding=1) * 0.1
    return F.conv2d(x, layers[-1][0], layers[-1][1], padding=1) + xs[-1]


def time_patches_per_s(flat, xs_np, num_layers, feature_size, budget_s=15.0, threads=None):
    """Run the graph on growing samples of xs_np until ~budget_s of CPU work; return (patches/s, sample, cores)."""
    if threads:
        torch.set_num_threads(threads)
    cores = torch.get_num_threads()
    cin = sum(a.shape[1] for a in xs_np)
    cout = xs_np[-1].shape[1]
    layers = build(flat, cin, cout, num_layers, feature_size)
    n = 4
    with torch.no_grad():
        xs = [torch.from_numpy(a[:n]) for a in xs_np]
        forward(layers, xs, num_layers)                       # warm-up (oneDNN primitive creation)
        done, spent = 0, 0.0
        while spent < budget_s and done < 10 * xs_np[0].shape[0]:
            xs = [torch.from_numpy(a[:n]) for a in xs_np]
            t0 = time.perf_counter()
            y = forward(layers, xs, num_layers)
            dt = time.perf_counter() - t0
            spent += dt
            done += xs[0].shape[0]
            if dt < 2.0 and n * 2 <= xs_np[0].shape[0]:
                n *= 2
    return done / spent, done, cores, y.numpy()

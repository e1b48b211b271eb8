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


def cpu_model_name():
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name'):
                return ln.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cgroup_cpu_limit():
    """CPUs this process may use according to its cgroup's CPU quota (v2 cpu.max, v1 cfs quota); None if unlimited."""
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            return max(1, int(float(q) / float(per) + 0.5))
    except (OSError, ValueError):
        pass
    try:
        q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        if q > 0:
            return max(1, int(q / float(per) + 0.5))
    except (OSError, ValueError):
        pass
    return None


def physical_cores():
    """Physical cores available to this process: SMT siblings counted once, affinity mask and cgroup CPU quota
    respected (a container limited to 16 CPUs of a 128-core host must not run 128 oneDNN threads)."""
    n = _physical_cores_affinity()
    lim = cgroup_cpu_limit()
    return min(n, lim) if lim else n


def _physical_cores_affinity():
    try:
        import os
        avail = os.sched_getaffinity(0)
        seen = set()
        for c in avail:
            try:
                sib = open('/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list' % c).read().strip()
            except OSError:
                sib = str(c)
            seen.add(sib)
        return max(1, len(seen))
    except Exception:
        return max(1, torch.get_num_threads())


def time_patches_per_s(flat, xs_np, num_layers, feature_size, budget_s=15.0, threads=None, batch=64):
    """The graph on a FIXED batch of `batch` patches of xs_np, warm-up excluded, repeated until ~budget_s of CPU
    work: (patches/s, patches timed, threads used, last output, GFLOP/s)."""
    torch.set_num_threads(int(threads) if threads else physical_cores())
    cores = torch.get_num_threads()
    cin = sum(a.shape[1] for a in xs_np)
    cout = xs_np[-1].shape[1]
    layers = build(flat, cin, cout, num_layers, feature_size)
    n = min(batch, xs_np[0].shape[0])
    h, w = xs_np[0].shape[2:]
    flop = 2.0 * 9 * (cin * feature_size + 2 * num_layers * feature_size ** 2 + feature_size * cout) * h * w * n
    with torch.no_grad():
        xs = [torch.from_numpy(a[:n]) for a in xs_np]
        forward(layers, xs, num_layers)                       # warm-up (oneDNN primitive creation, thread pool)
        forward(layers, xs, num_layers)
        done, spent = 0, 0.0
        while spent < budget_s:
            t0 = time.perf_counter()
            y = forward(layers, xs, num_layers)
            spent += time.perf_counter() - t0
            done += n
    return done / spent, done, cores, y.numpy(), flop * (done / n) / spent / 1e9

"""What the GPU box gives the CPU baseline: CPU count, affinity, cgroup CPU quota, torch threads, and the oneDNN graph's rate at several thread counts."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
print('nproc', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpuset.cpus.effective'):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, 'n/a')
from oracle import cpu_graph, dsen2_oracle as do
flat = do.he_uniform_weights(10,6,6,128,seed=1)
xs = do.synthetic_inputs(64,32,32,(4,6),seed=0)
for t in (16, 32, 64, 128):
    r = cpu_graph.time_patches_per_s(flat, xs, 6, 128, budget_s=4, threads=t)
    print('threads', t, 'patches/s %.1f' % r[0], 'GFLOP/s %.0f' % r[4])

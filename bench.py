#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on MI355X: 32x32 patches/s for DSen2_20 (d=6, F=128, fp32) at
batch 512 per GPU, synthetic inputs resident in HBM, random-init (he_uniform) weights.

    python bench.py [--gpus N] [--steps K] [--warmup W]
N>1 is launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
one rank per GPU over RCCL.  A step = one forward pass of 512 patches per rank (weak scaling: patches
are independent units, no collective inside the network); the gather of each step's outputs to rank 0
("gather of outputs over xGMI") is issued asynchronously, overlaps the next step's forward, and is
completed inside the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H = W = 32
# BASELINE.json configs.  The default (what the driver runs) is configs[1]; the others are for our own runs.
CONFIGS = {
    'dsen2_20_fp32': dict(metric='32x32x6 patches/sec (DSen2_20, d=6, batch 512)', bands=(4, 6), d=6, f=128,
                          batch=512, precision='fp32', dtype='f32', peak=157.3,
                          workload='DSen2_20 d=6 F=128 fp32'),
    'dsen2_60_fp32': dict(metric='32x32x2 patches/sec (DSen2_60, d=6, batch 512)', bands=(4, 6, 2), d=6, f=128,
                          batch=512, precision='fp32', dtype='f32', peak=157.3,
                          workload='DSen2_60 d=6 F=128 fp32 (12-band input, 2-band output)'),
    'vdsen2_20_fp32': dict(metric='32x32x6 patches/sec (VDSen2_20, d=32, F=256, batch 256, fp32)', bands=(4, 6), d=32,
                           f=256, batch=256, precision='fp32', dtype='f32', peak=157.3,
                           workload='VDSen2_20 d=32 F=256 fp32'),
    'dsen2_20_bf16': dict(metric='32x32x6 patches/sec (DSen2_20, d=6, F=128, batch 512, bf16)', bands=(4, 6), d=6, f=128,
                          batch=512, precision='bf16', dtype='bf16', peak=2500.0,
                          workload='DSen2_20 d=6 F=128, bf16 operands / fp32 accumulate + residual stream (not a BASELINE config)'),
    'dsen2_20_bf16x3': dict(metric='32x32x6 patches/sec (DSen2_20, d=6, F=128, batch 512, bf16x3)', bands=(4, 6), d=6, f=128,
                            batch=512, precision='bf16x3', dtype='bf16x3', peak=2500.0, mfmas_per_product=3,
                            workload='DSen2_20 d=6 F=128, bf16x3: every fp32 operand = two bf16 numbers, three bf16 MFMAs per product, '
                                     'fp32 accumulate + exact fp32 residual stream (<= 1e-4 RMSE mode; not a BASELINE config, never the headline)'),
    'vdsen2_20_bf16x3': dict(metric='32x32x6 patches/sec (VDSen2_20, d=32, F=256, batch 256, bf16x3)', bands=(4, 6), d=32, f=256,
                             batch=256, precision='bf16x3', dtype='bf16x3', peak=2500.0, mfmas_per_product=3,
                             workload='VDSen2_20 d=32 F=256, bf16x3 (three bf16 MFMAs per product, fp32 accumulate + exact fp32 residual stream)'),
    'vdsen2_20_bf16': dict(metric='32x32x6 patches/sec (VDSen2_20, d=32, F=256, batch 256, bf16)', bands=(4, 6), d=32,
                           f=256, batch=256, precision='bf16', dtype='bf16', peak=2500.0,
                           workload='VDSen2_20 d=32 F=256, bf16 operands / fp32 accumulate + residual stream'),
}
# peaks: MI355X_MICROARCH.md — FP32 matrix 157.3 TFLOP/s, BF16 MFMA ~2.5 PFLOP/s dense


def other_config(name, seconds, dev):
    """One of the non-headline configs for about `seconds` of GPU time: the same quantities the headline line carries, by the
    same means — a loop of plain forward passes between two events on the launch stream (value, ms_per_step), then
    dsen2_model_forward_profile's four-event passes for the dominant kernel's launch duration (roofline_frac)."""
    import torch
    from dsen2_amd import weights as dweights
    from dsen2_amd.DSen2Net import s2model
    cfg = CONFIGS[name]
    bands, d, f, n = cfg['bands'], cfg['d'], cfg['f'], cfg['batch']
    t_setup = time.perf_counter()
    model = s2model(tuple((b, None, None) for b in bands), num_layers=d, feature_size=f, device=dev, precision=cfg['precision'])
    model.set_weights_flat(dweights.random_he_uniform(sum(bands), bands[-1], d, f, seed=1))
    rng = np.random.Generator(np.random.PCG64(0))
    xs = [torch.from_numpy(rng.random((n, c, H, W), dtype=np.float32) * np.float32(5.0)).to(dev) for c in bands]
    out = torch.empty((n, bands[-1], H, W), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup
    t_pre = time.perf_counter()
    n_warm = 0
    while time.perf_counter() - t_pre < 0.060 or n_warm < 3:          # back to the steady clock after the set-up's idle stretch
        model.forward_device(xs, out=out)
        torch.cuda.synchronize()
        n_warm += 1
    est_ms = (time.perf_counter() - t_pre) * 1e3 / n_warm
    steps = max(3, int(0.6 * seconds * 1e3 / est_ms))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        model.forward_device(xs, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms_per_step = e0.elapsed_time(e1) / steps
    n_prof = max(3, int(0.4 * seconds * 1e3 / ms_per_step))
    prof = model.profile_forward(xs, out=out, iters=n_prof, warm=3)
    launches = model.body_launches(n, H, W)
    ms_launch = prof['body_ms'] / launches
    flop_launch = n * H * W * 2 * 9 * f * f * (2 * d // launches) * cfg.get('mfmas_per_product', 1)
    achieved = flop_launch / (ms_launch * 1e-3) / 1e12
    flop_net = n * H * W * 2 * 9 * (sum(bands) * f + 2 * d * f * f + f * bands[-1])
    res = {'metric': cfg['metric'], 'value': round(n / (ms_per_step * 1e-3), 1), 'unit': 'patches/s', 'ms_per_step': round(ms_per_step, 4),
           'steps': steps, 'dtype': cfg['dtype'], 'batch': n, 'workload': cfg['workload'],
           'net_tflops': round(flop_net / (ms_per_step * 1e-3) / 1e12, 2),
           'roofline_frac': round(achieved / cfg['peak'], 4), 'achieved_tflops': round(achieved, 2), 'peak_tflops': cfg['peak'],
           'launches_per_forward': launches, 'ms_per_launch': round(ms_launch, 4), 'profiled_passes': n_prof,
           'forward_ms': round(prof['forward_ms'], 4), 'first_ms': round(prof['first_ms'], 4), 'out_ms': round(prof['out_ms'], 4),
           'finite': bool(torch.isfinite(out).all().item()), 'setup_s': round(t_setup, 2)}
    del model, xs, out
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default='dsen2_20_fp32', choices=sorted(CONFIGS))
    ap.add_argument('--batch', type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-budget', type=float, default=15.0, help='seconds of CPU work for cpu_baseline')
    ap.add_argument('--roofline-seconds', type=float, default=1.5, help='GPU time of the instrumented passes behind `roofline` (right after the timed loop, same clock state), and again of their event-free replay')
    ap.add_argument('--sustain-seconds', type=float, default=5.0, help='then this many seconds of plain forward passes: `roofline.sustained_ms_per_step`, the rate the chip holds once it is thermally settled (the timed loop itself is 0.26 s) — also what lets an outside utilisation sampler see the card busy; 0 skips it')
    ap.add_argument('--other-seconds', type=float, default=2.0, help='after the headline (N = 1, default config only): this many seconds of GPU time for each of BASELINE configs[2] (DSen2_60 fp32) and configs[4] (VDSen2_20 bf16) -> `other_configs`; 0 skips them')
    ap.add_argument('--no-gather', action='store_true', help='skip the per-step output all-gather (N>1)')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='gloo = rehearsal of the N>1 control flow on a box with fewer GPUs than ranks (ranks share devices; '
                         'the number it prints is not a measurement)')
    args = ap.parse_args()

    cfg = CONFIGS[args.config]
    BANDS, NUM_LAYERS, FEAT = cfg['bands'], cfg['d'], cfg['f']
    BATCH = cfg['batch']
    if args.batch <= 0:
        args.batch = BATCH
    FLOP_PER_PIXEL_BODY = 2 * 9 * FEAT * FEAT                    # one 3x3xFxF conv (294 912 FLOP/px at F=128)
    FLOP_PER_PIXEL_NET = 2 * 9 * (sum(BANDS) * FEAT + 2 * NUM_LAYERS * FEAT * FEAT + FEAT * BANDS[-1])
    PEAK = cfg['peak']

    import torch
    import torch.distributed as td
    from dsen2_amd import dist as ddist
    from dsen2_amd import weights as dweights
    from dsen2_amd.DSen2Net import s2model

    rank, local_rank, world = ddist.launched_world()
    if world != args.gpus:
        if rank == 0:
            sys.stderr.write('bench.py: --gpus %d but WORLD_SIZE=%d; launch N>1 with torch.distributed.run\n'
                             % (args.gpus, world))
        sys.exit(2)
    # one process per GPU: this rank's device (LOCAL_RANK), the dmabuf-IPC environment RCCL needs (set before HIP
    # initialises) and the process group — the same entry the drop-in CLI uses (dsen2_amd/dist.py)
    # (at N > 1 it also makes first contact: group timeout 120 s instead of torch's 10 minutes, an all-reduce of 1 that must
    # give N — any failure names its step and ends the rank with a non-zero code, dsen2_amd/dist.py::guarded_step)
    rank, world, dev = ddist.init_from_env(args.backend)
    # ---- setup (untimed): weights on rank 0 -> RCCL broadcast; synthetic inputs in HBM ----
    n_params = dweights.num_params(sum(BANDS), BANDS[-1], NUM_LAYERS, FEAT)
    flat = dweights.random_he_uniform(sum(BANDS), BANDS[-1], NUM_LAYERS, FEAT, seed=1) if rank == 0 else None
    if world > 1:
        with ddist.guarded_step('broadcast of the weights (C1)'):
            flat = ddist.broadcast_weights(flat, n_params, device=dev)
        # from here on the whole N > 1 run has a deadline below the 600 s a driver allows: whatever hangs later (a gather
        # that never completes is first ended by RCCL's own watchdog at the group timeout) leaves every thread's stack on
        # stderr and a non-zero exit code.  (Armed after the guarded steps: they use the same one-shot watchdog.)
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ.get('DSEN2_BENCH_DEADLINE', '480')), exit=True)
    else:
        flat = ddist.broadcast_weights(flat, n_params, device=dev)
    model = s2model(tuple((b, None, None) for b in BANDS), num_layers=NUM_LAYERS, feature_size=FEAT, device=dev,
                    precision=cfg['precision'])
    model.set_weights_flat(flat)
    rng = np.random.Generator(np.random.PCG64(rank))             # U[0,1)*5, PCG64(seed)
    xs_np = [(rng.random((args.batch, c, H, W), dtype=np.float32) * np.float32(5.0)) for c in BANDS]
    xs = [torch.from_numpy(a).to(dev) for a in xs_np]
    do_gather = world > 1 and not args.no_gather
    cdev = dev if args.backend == 'nccl' else torch.device('cpu')          # gloo stages through host memory
    # two output buffers: the gather of step i (RCCL stream) overlaps the forward of step i+1 (compute stream)
    outs = [torch.empty((args.batch, BANDS[-1], H, W), dtype=torch.float32, device=dev) for _ in range(2)]
    out = outs[0]
    gather_list = None
    if do_gather and rank == 0:
        gather_list = [[torch.empty((args.batch, BANDS[-1], H, W), dtype=torch.float32, device=cdev) for _ in range(world)]
                       for _ in range(2)]
    pending = [None, None]
    # operands of the per-epilogue timings below, allocated NOW: an allocation of 3 x 256 MiB between the timed loop and the
    # instrumented passes idles the GPU for milliseconds, and it then needs ~25 ms to come back to its steady clock
    # (profiles/r04_ablation.md §2)
    bf = cfg['precision'] == 'bf16'
    rl_a = rl_r = rl_o = None
    if rank == 0:
        rl_a = torch.randn((args.batch, H, W, FEAT), dtype=torch.float32, device=dev)
        rl_r = torch.randn((args.batch, H, W, FEAT), dtype=torch.float32, device=dev)
        rl_o = torch.empty_like(rl_a)
        if bf:
            rl_a = rl_a.to(torch.bfloat16)      # conv-A writes bf16 into `o`; conv-B updates `r` in place, read as the (hi, lo) planes

    # What the gather costs the compute stream (N > 1 only; nothing of this runs at N = 1).  With RCCL `wait()` does not
    # block the host: it makes the compute stream wait for the collective — so the stall is measured there, between an
    # event recorded right before the wait and one right after it (both fire at once unless the gather is still running when
    # the previous forward has finished).  The host-side seconds inside wait() are kept too: that IS the stall under gloo.
    gw = {'host_s': 0.0, 'events': [], 'on': False}

    def wait_gather(b):
        if pending[b] is None:
            return
        if gw['on']:
            if args.backend == 'nccl':
                ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ea.record()
            t = time.perf_counter()
            pending[b].wait()
            gw['host_s'] += time.perf_counter() - t
            if args.backend == 'nccl':
                eb.record()
                gw['events'].append((ea, eb))
        else:
            pending[b].wait()
        pending[b] = None

    def step(i, gather=do_gather):
        """forward of 512 patches into outs[i%2]; then start gathering it to rank 0 (7 peers x 12.6 MB, one xGMI
        link each) without waiting: the handle is waited on before outs[i%2] is written again, two steps later."""
        b = i & 1
        wait_gather(b)
        model.forward_device(xs, out=outs[b])
        if gather:
            src = outs[b] if args.backend == 'nccl' else outs[b].cpu()
            pending[b] = td.gather(src, gather_list[b] if rank == 0 else None, dst=0, async_op=True)

    def drain():
        for b in range(2):
            wait_gather(b)

    # Untimed, before the W warm-up steps: at least ~40 ms of forwards.  After an idle stretch (process start, weight upload) the
    # chip needs ~25 ms of work to reach its steady clock (profiles/r04_ablation.md §2); W steps of a SHORT step (bf16: 1.8 ms)
    # would leave the timed loop inside that ramp.  The headline config's W x 12.8 ms covers it anyway.
    torch.cuda.synchronize()
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.040:
        model.forward_device(xs, out=outs[0])
        torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    drain()
    torch.cuda.synchronize()

    def timed_steps(gather):
        """EXACTLY K steps between barrier + synchronize on both sides.  Returns (seconds incl. the closing barrier, this
        rank's own seconds up to its last kernel and gather, before that barrier)."""
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i, gather)
        drain()
        torch.cuda.synchronize()
        own = time.perf_counter() - t0
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, own

    gw['on'] = do_gather
    elapsed, own_s = timed_steps(do_gather)
    gw['on'] = False
    diag = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        elapsed = float(t.item())
        # ---- after the timed region: what makes an N > 1 line explain itself ----
        # (a) every rank's own step, and how long its compute stream stood waiting for a gather
        stall_ms = sum(ea.elapsed_time(eb) for ea, eb in gw['events'])
        # (b) the control: the same K steps, same bracket, with the gather OFF — compute alone on all N GPUs at once
        elapsed_ng, own_ng = timed_steps(False)
        mine = torch.tensor([own_s, stall_ms * 1e-3, gw['host_s'], elapsed_ng, own_ng], dtype=torch.float64, device=cdev)
        every = [torch.empty_like(mine) for _ in range(world)]
        td.all_gather(every, mine)
        every = np.array([e.cpu().numpy() for e in every]) / args.steps * 1e3            # [rank, quantity] in ms per step
        diag = every
    out = outs[(args.steps - 1) & 1]

    ms_per_step = elapsed / args.steps * 1e3
    value = world * args.batch * args.steps / elapsed

    result = {
        'metric': cfg['metric'],
        'value': round(value, 1), 'unit': 'patches/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': cfg['dtype'], 'data': 'synthetic',
        'config': {'workload': '%s, %d synthetic 32x32 patches (%s bands) per GPU per step, he_uniform random-init '
                               'weights' % (cfg['workload'], args.batch, '+'.join(str(b) for b in BANDS)),
                   'batch_per_gpu': args.batch, 'patch': [H, W], 'parallelism': 'patch-sharded dp%d' % world, 'backend': 'rccl' if args.backend == 'nccl' else 'gloo-rehearsal',
                   'output_gather': bool(do_gather)},
        'net_tflops': round(value * H * W * FLOP_PER_PIXEL_NET / 1e12, 2),
    }
    if world > 1:
        fc = ddist.first_contact()
        r4 = lambda col: [round(float(v), 4) for v in diag[:, col]]      # noqa: E731
        result.update({
            # all-reduce of 1 over the group at start-up: the collectives really spanned N ranks
            'ranks_in_collective': fc.get('ranks_in_collective'),
            'dist_setup_s': fc.get('seconds'),
            # each rank's own K steps (its kernels + its gathers, before the closing barrier) — `ms_per_step` is the slowest
            # rank's incl. that barrier
            'per_rank_ms_per_step': {'min': round(float(diag[:, 0].min()), 4), 'max': round(float(diag[:, 0].max()), 4), 'ranks': r4(0)},
            # time the compute stream stood in `pending[b].wait()` for a gather still in flight (HIP events around the wait;
            # under gloo the host blocks instead: `host`), per step, per rank — rank 0 receives, the others send
            'gather_wait_ms_per_step': {'max': round(float(diag[:, 1].max()), 4), 'ranks': r4(1), 'host_max': round(float(diag[:, 2].max()), 4)} if do_gather else None,
            # the control, same run, same bracket, gather off: what N GPUs computing side by side do without RCCL's kernels
            # competing with two CU-filling persistent kernels.  ms_per_step - ms_per_step_no_gather = what the gather costs.
            'ms_per_step_no_gather': round(float(diag[:, 3].max()), 4),
            'per_rank_ms_per_step_no_gather': {'min': round(float(diag[:, 4].min()), 4), 'max': round(float(diag[:, 4].max()), 4)},
            'value_no_gather': round(world * args.batch / (float(diag[:, 3].max()) * 1e-3), 1),
        })

    if rank == 0:
        # ---- roofline of the dominant kernel: the 3x3x128x128 body convolution (98.97 % of FLOPs) ----
        pix = args.batch * H * W
        a, r, o = rl_a, rl_r, rl_o
        # (1) the launch duration inside the running network: `steps` more forward passes on the bench inputs with FOUR HIP
        #     events each on the launch stream (dsen2_model_forward_profile): whole forward, first convolution, the 2d body
        #     convolutions, output convolution — consecutive intervals between the same time stamps, so
        #     launches x ms_per_launch + first_ms + out_ms = forward_ms by construction
        #     (3 plain passes are enqueued right before them, no synchronisation in between: the GPU is at its steady clock)
        # at least `steps` instrumented passes, as many as fit --roofline-seconds (1.5 s: ~100 at the default config): a stable mean
        n_prof = max(args.steps, min(int(100 * args.roofline_seconds / 1.5), int(1e3 * args.roofline_seconds / max(ms_per_step, 1e-3))))
        prof = model.profile_forward(xs, out=outs[0], iters=n_prof, warm=3)
        ms = prof['body_ms'] / (2 * NUM_LAYERS)
        # (1b) the same passes WITHOUT the events inside them, right after and equally warm (the kernels run on torch's
        #     current stream, so two torch events bracket them): what the four event records per forward cost, and whether
        #     the chip runs the timed loop's step again
        for _ in range(3):
            model.forward_device(xs, out=outs[0])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n_prof):
            model.forward_device(xs, out=outs[0])
        e1.record()
        torch.cuda.synchronize()
        replay_ms = e0.elapsed_time(e1) / n_prof
        # (1c) sustained: several seconds of the same passes, back to back after the above
        sustained_ms, n_sus = None, 0
        if args.sustain_seconds > 0:
            n_sus = max(1, int(1e3 * args.sustain_seconds / max(ms_per_step, 1e-3)))
            e0.record()
            for _ in range(n_sus):
                model.forward_device(xs, out=outs[0])
            e1.record()
            torch.cuda.synchronize()
            sustained_ms = e0.elapsed_time(e1) / n_sus
        # (2) each epilogue alone, back to back, on dense random operands (no ReLU zeros: the chip clocks lower)
        x3 = cfg['precision'] == 'bf16x3'
        ms_relu = ms_res = None
        if not x3:
            ms_relu = model.time_body_conv(1, a, None, o, iters=10)          # conv-A (+bias+ReLU)
            ms_res = model.time_body_conv(2, a, r, o, iters=10)              # conv-B (+bias, *0.1, +residual)
        # bf16x3: a product is three bf16 MFMAs (hi*hi + hi*lo + lo*hi) — the matrix pipe's work and the roofline fraction
        # count all of them against the bf16 peak; `algorithmic_tflops` is the convolution's own 2*9*F*F per pixel
        flops = pix * FLOP_PER_PIXEL_BODY * cfg.get('mfmas_per_product', 1)
        # the dominant kernel: one launch per body convolution, or (bf16, a batch of whole patches per CU) ONE chain
        # launch over all 2d of them — then a launch's work and duration are 2d layers'
        launches = model.body_launches(args.batch, H, W)
        per_launch = 2 * NUM_LAYERS // launches
        # HBM traffic of that kernel: PMC counters cannot be read in-process, so this is the committed profile of this
        # config (tools/profile_round.sh -> tools/update_traffic_json.py) — quoted only while the instruction stream it was
        # measured on is the one this library was built from (dsen2_amd/kernel_isa.json, written by the build)
        traffic, traffic_src, tdat_ok = None, 'no committed PMC profile of this config', None
        tj = os.path.join(ROOT, 'profiles', 'body_conv_traffic.json')
        ij = os.path.join(ROOT, 'dsen2_amd', 'kernel_isa.json')
        if os.path.exists(tj) and args.batch == BATCH:
            # (DSen2_60's residual blocks run the very kernel of DSen2_20 on the same 512 x 32 x 32 x 128 tensors)
            tkey = {'dsen2_60_fp32': 'dsen2_20_fp32'}.get(args.config, args.config)
            tdat = json.load(open(tj)).get(tkey)
            built = json.load(open(ij)).get(tkey, {}).get('isa_sha256') if os.path.exists(ij) else None
            if tdat and built and tdat.get('isa_sha256') == built:
                traffic, traffic_src, tdat_ok = tdat['traffic_bytes'], tdat['source'], tdat
            elif tdat:
                traffic_src = 'STALE, not quoted: profiles/body_conv_traffic.json was measured on kernel ISA %s, this build is %s' % (
                    str(tdat.get('isa_sha256'))[:12], str(built)[:12])
        achieved = flops / (ms * 1e-3) / 1e12
        closure = launches * ms * per_launch + prof['first_ms'] + prof['out_ms']
        result['roofline'] = {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': PEAK,
                              'unit': 'TFLOP/s', 'frac': round(achieved / PEAK, 4), 'traffic': traffic,
                              'traffic_unit': 'bytes/launch (PMC, separate rocprofv3 passes; see traffic_source)',
                              'traffic_source': traffic_src,
                              # north_star: "rocprof-reported MFMA utilisation and HBM GB/s against gfx950 peak" — the PMC traffic over
                              # THIS run's launch duration against HBM3E's 8 TB/s, and the matrix pipes' busy fraction from the same
                              # committed PMC passes (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs))
                              'hbm_gbps': round(traffic / (ms * per_launch * 1e-3) / 1e9, 1) if traffic else None,
                              'hbm_frac_of_8TBps': round(traffic / (ms * per_launch * 1e-3) / 8e12, 4) if traffic else None,
                              'algorithmic_bytes': tdat_ok.get('algorithmic_bytes') if tdat_ok else None,
                              'mfma_busy_pmc': tdat_ok.get('mfma_busy') if tdat_ok else None,
                              'kernel': '%s (3x3x%dx%d, %s, persistent%s)' % (
                                  ('conv3x3_body16w_x3_chain_kernel' if launches == 1 else 'conv3x3_body16w_x3_kernel') if x3 else ('conv3x3_body16w_chain_kernel' if launches == 1 else 'conv3x3_body16w_kernel') if bf else 'conv3x3_body32_kernel',
                                  FEAT, FEAT, 'three bf16 MFMA 16x16x32 per product (hi*hi + hi*lo + lo*hi), LDS-DMA staging, 16x32-pixel items' if x3
                                  else 'bf16 MFMA 16x16x32, LDS-DMA staging, 16x32-pixel items' if bf else 'fp32 MFMA 32x32x2, LDS-DMA staging',
                                  '; ONE launch over all %d body convolutions, a workgroup owns its patches through every layer' % (2 * NUM_LAYERS) if launches == 1 else ''),
                              'ms_per_launch': round(ms * per_launch, 4),
                              'ms_per_launch_source': 'HIP events on the launch stream around the %d body-conv launch%s of each of %d forward passes (4 events per pass)' % (launches, '' if launches == 1 else 'es', n_prof),
                              'launches_per_forward': launches, 'convolutions_per_launch': per_launch, 'ms_per_convolution': round(ms, 4),
                              # the line closes on itself: launches x ms_per_launch + first_ms + out_ms = forward_ms (same events)
                              'forward_ms': round(prof['forward_ms'], 4), 'first_ms': round(prof['first_ms'], 4),
                              'out_ms': round(prof['out_ms'], 4), 'closure_ms': round(closure, 4),
                              # ... and against the un-instrumented step: forward_period_ms = one instrumented pass with
                              # its four event records and the gap to the next pass; replay_ms_per_step = the same passes
                              # without those events, right after, equally warm; event_cost_ms = their difference;
                              # replay_vs_timed_loop_ms = replay minus the timed loop's ms_per_step (which also holds the
                              # bracketing synchronisations and, at N > 1, the gather): ~0 when the chip runs the same clock
                              'forward_period_ms': round(prof['wall_ms'], 4), 'replay_ms_per_step': round(replay_ms, 4),
                              'event_cost_ms': round(prof['wall_ms'] - replay_ms, 4),
                              'replay_vs_timed_loop_ms': round(replay_ms - ms_per_step, 4),
                              # the step the chip holds over --sustain-seconds of uninterrupted passes (thermally settled; the
                              # figures above are from the first ~3 s after start-up)
                              'sustained_ms_per_step': round(sustained_ms, 4) if sustained_ms is not None else None,
                              'sustained_passes': n_sus,
                              'ms_relu_randn': round(ms_relu, 4) if ms_relu is not None else None,
                              'ms_residual_randn': round(ms_res, 4) if ms_res is not None else None,
                              'flop_per_launch': flops * per_launch,
                              'algorithmic_tflops': round(pix * FLOP_PER_PIXEL_BODY / (ms * 1e-3) / 1e12, 2)}
        del a, r, o
        rl_a = rl_r = rl_o = None

    if rank == 0 and world == 1 and args.config == 'dsen2_20_fp32' and args.other_seconds > 0:
        # ---- the other single-GPU BASELINE configs, on the same box, AFTER everything the headline needs (its timed loop and
        # roofline legs are over; the chip is warm) — configs[2] DSen2_60 fp32 and configs[4] VDSen2_20 bf16, ~args.other_seconds
        # of GPU time each.  Not the headline, not `value`: they put two more BASELINE configs on the driver's record.
        model.release_workspaces()
        result['other_configs'] = {}
        for name in ('dsen2_60_fp32', 'vdsen2_20_bf16'):
            try:
                result['other_configs'][name] = other_config(name, args.other_seconds, dev)
            except Exception as e:       # noqa: BLE001 — a rider must never cost the headline its line: reported in its place
                result['other_configs'][name] = {'error': '%s: %s' % (type(e).__name__, e)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config == 'dsen2_20_fp32':
        # ---- CPU baseline: the same graph on the host cores, bounded sample (oracle/ = checker code) ----
        from oracle import cpu_graph
        pps, sample, cores, y_cpu, gflops = cpu_graph.time_patches_per_s(flat, xs_np, NUM_LAYERS, FEAT, budget_s=args.cpu_budget)
        result['cpu_baseline'] = {'value': round(pps, 2), 'unit': 'patches/s', 'cores': cores, 'kind': 'port',
                                  'gflops': round(gflops, 1), 'cpu': cpu_graph.cpu_model_name(),
                                  'sample': 'the first 64 patches of the same synthetic batch, run repeatedly as ONE fixed '
                                            'batch (%d patches timed in ~%.0f s, two warm-up passes excluded); '
                                            'torch-CPU fp32 conv2d graph (oneDNN), one thread per physical core'
                                            % (sample, args.cpu_budget)}
        n = y_cpu.shape[0]
        diff = out[:n].cpu().numpy().astype(np.float64) - y_cpu.astype(np.float64)
        result['rmse_vs_cpu_fp32'] = float(np.sqrt(np.mean(diff * diff)))

    if rank == 0:
        print(json.dumps(result), flush=True)
    ddist.finalize()


if __name__ == '__main__':
    main()

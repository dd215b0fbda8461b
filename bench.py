#!/usr/bin/env python
"""bench.py — utterances/sec of the Nabu training step on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N
          --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...)

Workload (BASELINE.json configs[1]/[3], "cfg2"): 4-layer Listener-512 (3 pyramidal
BLSTM layers + 1 BLSTM) + DNNDecoder + CTC, batch 32 x 1000 frames x 40 fbank per
GPU, fp32, synthetic seeded data, random-init weights.  One step = forward + CTC
loss/gradient + backward + per-element clip + Adam (+ RCCL all-reduce of the
clipped gradients when N > 1, weak scaling: every rank has its own batch of 32).
Inputs are resident in HBM before the timed region.

One JSON line is printed by rank 0.  Extra objects:
  roofline     — the dominant kernel (the recurrent LSTM step), ALGORITHMIC bytes
                 (per timestep per direction W_h + x-projection +
                 gate activations + h/c state) / duration measured with HIP events
                 recorded by the library around the recurrent launches, inside the
                 timed region, on the launch stream.
  cpu_baseline — the oracle (NumPy float32 restatement of the reference graph at
                 TF op granularity, "port") timed on the host cores on a bounded
                 sample.  The reference's own TF-1.8 trainer cannot run here.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

B, T, D, H, C = 32, 1000, 40, 512, 40
LAYER_T = (1000, 500, 250, 125)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def step_bytes(batch, hidden):
    """algorithmic bytes of ONE recurrent timestep of ONE direction"""
    return 4 * (hidden * 4 * hidden + batch * 4 * hidden + batch * 4 * hidden + 4 * batch * hidden)


def cpu_baseline(seconds_budget=30.0):
    """Time the oracle (float32) for one full training step of the cfg2 model on a
    bounded sample: all 32 utterances, the first T_s frames (cost is linear in T)."""
    import numpy as np
    from oracle import nabu_oracle as O
    from nabu_amd.processing.synthetic import SyntheticData
    Ts = 200
    rng = np.random.default_rng(99)
    dims = [D, 4 * H, 4 * H, 4 * H]
    layers = [{k: O.glorot_uniform(rng, (d + H, 4 * H)) if 'kernel' in k else O.glorot_uniform(rng, (4 * H,))
               for k in ('fw_kernel', 'fw_bias', 'bw_kernel', 'bw_bias')} for d in dims]
    W = O.glorot_uniform(rng, (2 * H, C)); bo = np.zeros(C, np.float32)
    data = SyntheticData(B, Ts, D, min_frames=Ts, min_labels=4, max_labels=12, time_reduction=8, seed=2234)
    batch = data.batch(0)
    x = batch['inputs']['features']

    def one_step():
        enc, el, caches = O.listener_fwd(x, batch['input_seq_length']['features'], layers)
        lg = O.linear_fwd(enc, W, bo)
        nll, dlg = O.ctc_loss(lg, el, batch['targets']['text'], batch['target_seq_length']['text'])
        de, dW, db = O.linear_bwd((dlg / B).astype(np.float32), enc, W)
        _, grads = O.listener_bwd(de, caches)
        for l, g in zip(layers, grads):
            for k in l:
                l[k], _, _ = O.clip_adam_update(l[k], g[k], np.zeros_like(l[k]), np.zeros_like(l[k]), 1, 1e-3)
        return float(nll.mean())
    t0 = time.time()
    one_step()
    dt = time.time() - t0
    reps = 1
    if dt * 3 < seconds_budget:          # a second repetition if it is cheap
        t0 = time.time()
        one_step()
        dt = min(dt, time.time() - t0)
        reps = 2
    utt_s = B / (dt * (T / float(Ts)))
    return {'value': round(utt_s, 3), 'unit': 'utterances/sec', 'cores': os.cpu_count(), 'kind': 'port',
            'sample': '%d step(s) of the cfg2 model, 32 utterances x first %d of 1000 frames, NumPy float32 '
                      'oracle at TF op granularity (one [B,in+H]x[in+H,4H] matmul per timestep per '
                      'direction), BLAS threads = all host cores; utt/s scaled by %d/1000 (cost linear in T); '
                      'the reference TF-1.8 trainer itself cannot run here' % (reps, Ts, Ts)}


def gemm_roofline(B, T, D, H, precision):
    """Second roofline object: the dense products of one cfg2 step (input projections, dz·Wxᵀ, xᵀ·dz,
    hᵀ·dz of every layer, both directions) launched back to back on the current stream and timed
    with events — MFMA-bound, fp32 peak from MI355X_MICROARCH.md."""
    import torch
    from nabu_amd import ops
    shapes = []
    Dl = D
    for l in range(4):
        BT = B * (T >> l)
        # the calls nabu_blstm_fwd / _bwd make, one set per direction (lstm.hip)
        shapes += [(0, 0, BT, 4 * H, Dl)] * 2                    # gates_d = x · Wx_d
        if l:
            shapes += [(0, 1, BT, Dl, 4 * H)] * 2                # dx (+)= dz_d · Wx_d^T
        shapes += [(1, 0, Dl, 4 * H, BT)] * 2 + [(1, 0, H, 4 * H, BT)] * 2     # dWx_d, dWh_d
        Dl = 4 * H
    bufs = {}
    def buf(n):
        if n not in bufs:
            bufs[n] = torch.randn(n, device='cuda')
        return bufs[n]
    calls = []
    for ta, tb, M, N, K in shapes:
        a = buf(M * K).view((K, M) if ta else (M, K))
        b = buf(K * N).view((N, K) if tb else (K, N))
        c = torch.empty((M, N), device='cuda')
        calls.append((a, b, c, bool(ta), bool(tb)))
    def run():
        for a, b, c, ta, tb in calls:
            ops.gemm(a, b, c, trans_a=ta, trans_b=tb, precision=precision)
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = sum(2.0 * M * N * K for _, _, M, N, K in shapes)
    tf = flops / ms / 1e9
    return {'bound': 'mfma', 'achieved': round(tf, 1), 'peak': 157.3, 'unit': 'TFLOP/s', 'frac': round(tf / 157.3, 4),
            'traffic': None, 'kernel': 'gemm_f32_fast_kernel<*> (+ split-K reduce)', 'ms_per_step': round(ms, 3),
            'flops_per_step': int(flops),
            'note': 'every dense product of a cfg2 step at its real shape, back to back, timed with events on the '
                    'launch stream; peak = dense fp32 MFMA (v_mfma_f32_32x32x2_f32) at 2.4 GHz'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--mode', default='auto', choices=['auto', 'stepwise', 'persistent'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-alt', action='store_true', help='skip the informational bf16x6 measurement')
    ap.add_argument('--no-gemm-roofline', action='store_true',
                    help='skip the roofline_gemm measurement (keeps a kernel trace of this command to the training steps)')
    ap.add_argument('--gemm-precision', default='f32', choices=['f32', 'bf16x6', 'bf16x3', 'bf16'],
                    help='arithmetic of the dense products (include/nabu_hip.h nabu_gemm_ex); the BASELINE metric '
                         'is fp32 = the default; the others are reported as such in config.gemm_arith')
    ap.add_argument('--workload', default='cfg2', choices=['cfg1', 'cfg2', 'cfg3', 'cfg5'],
                    help='cfg2 (default) is the BASELINE.json metric; cfg3 = same encoder + Speller; cfg5 = '
                         'location-aware LAS, batch 64x1600x80, bf16 input GEMMs (BASELINE.json configs[2]/[4]), '
                         'for information')
    args = ap.parse_args()

    import torch
    from nabu_amd import recipes, ops, _hip
    from nabu_amd.computing import dist
    from nabu_amd.neuralnetworks.components import layer
    from nabu_amd.neuralnetworks.trainers import trainer_factory, loss_functions
    from nabu_amd.processing.synthetic import SyntheticData

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    server = dist.create_server()
    rank, world = server.rank, server.world_size
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if world == 1:
        torch.cuda.set_device(0)
    _hip.lib()
    ops.set_gemm_precision(args.gemm_precision)
    layer.LSTM_MODE[0] = {'auto': ops.LSTM_AUTO, 'stepwise': ops.LSTM_STEPWISE,
                          'persistent': ops.LSTM_PERSISTENT}[args.mode]

    global B, T, D, H
    layer_t = None
    if args.workload == 'cfg1':
        # BASELINE.json configs[0]: DBLSTM 2 x 256 + CTC, 8 x 200 x 40 (the reference's CPU-runnable case)
        B, T, D, H = 8, 200, 40, 256
        layer_t = [T, T]
        mc, tc, ec = recipes.load_recipe('cfg1_dblstm_ctc')
        data = SyntheticData(B, T, D, min_frames=T, min_labels=10, max_labels=40, seed=1234 + rank)
    elif args.workload == 'cfg5':
        B, T, D = 64, 1600, 80
        mc, tc, ec = recipes.load_recipe('cfg5_las_location')
        data = SyntheticData(B, T, D, min_frames=T, min_labels=40, max_labels=159, eos=True, time_reduction=8,
                             seed=5234 + rank)
    elif args.workload == 'cfg3':
        mc, tc, ec = recipes.load_recipe('cfg3_las_vanilla')
        data = SyntheticData(B, T, D, min_frames=T, min_labels=20, max_labels=79, eos=True, time_reduction=8,
                             seed=3234 + rank)
    else:
        mc, tc, ec = recipes.load_recipe('cfg2_listener_ctc')
        data = SyntheticData(B, T, D, min_frames=T, min_labels=20, max_labels=60, time_reduction=8,
                             seed=4234 + rank)
    tr = trainer_factory.factory('standard')(conf=tc, dataconf=data, modelconf=mc, evaluatorconf=ec,
                                             expdir=None, server=server, task_index=rank)
    batches = [tr.to_device(data.batch(i)) for i in range(2)]      # resident in HBM
    # the event profiler is armed during the warm-up as well: its first use (event pool creation
    # inside the HIP runtime) stalls the queue for tens of milliseconds once
    prof = ops.enable_profiler()
    for i in range(max(args.warmup, 1)):
        tr.step(batches[i % 2])
    loss_functions.check_status()
    torch.cuda.synchronize()
    prof.collect()                                                 # drop the warm-up records
    server.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = tr.step(batches[i % 2])
    torch.cuda.synchronize()
    server.barrier()
    dt = time.perf_counter() - t0
    prof.enabled = False
    recs = prof.collect()
    final_loss = float(loss.item())
    loss_functions.check_status()
    alt = None
    if args.gemm_precision == 'f32' and args.workload == 'cfg2' and not args.no_alt:
        alt = alt_gemm_arith(tr, batches, server, min(args.steps, 5))
    tmax = torch.tensor([dt, alt[0] if alt else 0.0], dtype=torch.float64, device='cuda')
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax[0].item())
    if alt:
        alt = (float(tmax[1].item()), alt[1])
    if rank != 0:
        return

    # roofline of the dominant kernel: recurrent step(s), per launch
    import ctypes
    desc = _hip.BlstmDesc(ctypes.sizeof(_hip.BlstmDesc), B, T, D, H, T, layer.LSTM_MODE[0], 0)
    persistent = bool(_hip.lib().nabu_blstm_uses_persistent(ctypes.byref(desc)))
    tot_ms = sum(r[4] for r in recs)
    tot_steps = sum(r[2] for r in recs)                   # timesteps covered (both directions each)
    tot_bytes = sum(2 * r[2] * step_bytes(r[1], r[3]) for r in recs)
    launches = len(recs) if persistent else tot_steps
    per_launch_bytes = tot_bytes / max(launches, 1)
    per_launch_s = tot_ms * 1e-3 / max(launches, 1)
    achieved = per_launch_bytes / per_launch_s / 1e9 if per_launch_s > 0 else 0.0
    # HBM traffic per launch of the same kernels from the PMC passes (rocprofv3 --pmc FETCH_SIZE /
    # WRITE_SIZE, separate runs of this command; summarised by tools/pmc_summary.py with the
    # gfx950 corrections of MI355X_MICROARCH.md).  null when this workload was not profiled.
    traffic = None
    pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r01_cfg2_pmc_traffic.json')
    if persistent and args.workload == 'cfg2' and os.path.exists(pmc):
        with open(pmc) as fid:
            traffic = int(json.load(fid)['lstm_persist_traffic_bytes_per_launch'])
    roofline = {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic,
                'kernel': 'lstm_persist_{fwd,bwd}' if persistent else 'lstm_step_{fwd,bwd}_kernel',
                'bytes_per_launch': int(per_launch_bytes), 'us_per_launch': round(per_launch_s * 1e6, 3),
                'launches_timed': launches,
                'recurrent_ms_per_step': round(tot_ms / args.steps, 3),
                'note': 'algorithmic bytes (W_h streamed per timestep model, SURVEY.md 8(d)); '
                        'events recorded by the library around the recurrent launches'}
    if layer_t is None:
        layer_t = [T >> i for i in range(4)]
    step_bytes_total = 2 * 2 * sum(layer_t) * step_bytes(B, H)
    out = {
        'metric': {'cfg1': 'utterances/sec training step, 2x256 DBLSTM+CTC, batch 8x200x40 fbank',
                   'cfg2': 'utterances/sec training step, 4x512 Listener+CTC, batch 32x1000x40 fbank',
                   'cfg3': 'utterances/sec training step, Listener-512 + Speller (vanilla attention), batch 32x1000x40',
                   'cfg5': 'utterances/sec training step, Listener-512 + Speller (location-aware attention), '
                           'bf16 input GEMMs, batch 64x1600x80'}[args.workload],
        'value': round(world * B * args.steps / dt, 2), 'unit': 'utterances/sec', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32' if args.gemm_precision in ('f32', 'bf16x6') else 'f32 state / %s products' % args.gemm_precision,
        'data': 'synthetic',
        'config': {'workload': {'cfg1': 'cfg1: DBLSTM 2 x 256, DNNDecoder, CTC, Adam+clip; 8 utt x 200 frames x 40 fbank per GPU',
                                'cfg2': 'cfg2: Listener 3 pyramidal + 1 BLSTM x512, DNNDecoder, CTC, Adam+clip; '
                                        '32 utt x 1000 frames x 40 fbank per GPU',
                                'cfg3': 'cfg3: cfg2 encoder + Speller (1x512 LSTMCell, Bahdanau attention), '
                                        'average cross-entropy; 32 utt x 1000 frames x 40 fbank per GPU',
                                'cfg5': 'cfg5: Listener-512 (bf16 input GEMMs) + Speller (location-aware attention); '
                                        '64 utt x 1600 frames x 80 fbank per GPU'}[args.workload],
                   'global_batch': world * B, 'frames': T, 'parallelism': 'dp%d' % world,
                   'recurrent_path': 'persistent' if persistent else 'stepwise',
                   'gemm_arith': {'f32': 'f32 (v_mfma_f32_32x32x2_f32, exact fp32)',
                                  'bf16x6': 'f32 operands split into 3 bf16 pieces, 6 bf16 MFMA products, f32 accumulate',
                                  'bf16x3': 'f32 operands split into 2 bf16 pieces, 3 bf16 MFMA products, f32 accumulate',
                                  'bf16': 'operands rounded to bf16, f32 accumulate'}[args.gemm_precision]},
        'roofline': roofline,
        'roofline_gemm': (gemm_roofline(B, T, D, H, args.gemm_precision)
                          if args.workload == 'cfg2' and args.gemm_precision == 'f32' and not args.no_gemm_roofline
                          else None),
        'hbm_roofline_frac_whole_step': round(step_bytes_total / (dt / args.steps) / (HBM_PEAK_GBS * 1e9), 4),
        'final_loss': round(final_loss, 4),
    }
    if alt:
        n = min(args.steps, 5)
        out['alt_gemm_bf16x6'] = {
            'note': 'informational, not the headline: identical step with every dense product computed as 6 bf16 '
                    'MFMA products of 3-way split fp32 operands (fp32 accumulate; error vs float64 equal to the '
                    'exact-fp32 MFMA kernel, tests/test_hip_ops.py::test_gemm_bf16_split_precisions)',
            'value': round(world * B * n / alt[0], 2), 'ms_per_step': round(alt[0] / n * 1e3, 3), 'steps': n,
            'final_loss': round(alt[1], 4)}
    if world == 1 and not args.no_cpu_baseline and args.workload == 'cfg2':
        out['cpu_baseline'] = cpu_baseline()
    print(json.dumps(out))


def alt_gemm_arith(tr, batches, server, steps):
    '''informational: the same step with the dense products on the bf16 matrix pipe as 6 split
    products (fp32-level accuracy, tests/test_hip_ops.py); NOT the headline value'''
    import torch
    from nabu_amd import ops
    ops.set_gemm_precision('bf16x6')
    try:
        tr.step(batches[0])
        torch.cuda.synchronize()
        server.barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            loss = tr.step(batches[i % 2])
        torch.cuda.synchronize()
        server.barrier()
        dt = time.perf_counter() - t0
    finally:
        ops.set_gemm_precision('f32')
    return dt, float(loss.item())


if __name__ == '__main__':
    main()

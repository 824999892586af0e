#!/usr/bin/env python
"""Headline benchmark: query-correspondences/sec of the COTR forward path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json configs[1], SURVEY.md 8d "primary"): one 256x512 side-by-side pair,
1000 queries, zoom disabled -> one ``model(img[1,3,256,512], q[1,1000,2])`` call per step, per GPU.
Synthetic data and seeded random weights of the COTR architecture (no checkpoint exists offline).
Inputs are resident in HBM before the timed region.  N > 1 (launched by torch.distributed.run, one
rank per GPU, RCCL): every rank runs its own pair(s) (weak scaling), the predicted (x,y) of every
step are all-gathered over xGMI on RCCL's stream, overlapped with the next step.

Prints ONE JSON line on rank 0.  ``roofline`` is for the whole forward launch sequence (one "launch"
= one cotr_forward = the ~150 kernels of one step): achieved = FLOP(B,Q) / mean step time measured
with HIP events on the launch stream, against the fp32 MFMA peak (parity forces fp32 operands).
``cpu_baseline`` is the CPU oracle (a torch-CPU restatement of the reference, kind "port") timed on
this box's host cores on a bounded number of the same forward calls - rank 0, N == 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PAIRS_PER_GPU = 1
QUERIES = 1000
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, 2.4 GHz x 256 CUs
HBM_PEAK_GBS = 8000.0


def flop(b, q):
    """Algorithmic work of the path, SURVEY.md 2.2 / 8(d): per pair 24.641 GFLOP (backbone, input_proj,
    encoder, decoder K/V), per query 11.273 MFLOP (6 decoder layers + final norm + corr MLP once)."""
    return b * 24.641e9 + b * q * 11.273e6


def min_hbm_bytes(b, q):
    """Minimum HBM traffic per call: weights once + image + queries in / predictions out."""
    return 73.8e6 + b * (1.573e6 + 16 * q)


def usable_cores():
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota (the GPU
    box reports 256 CPUs but grants 16; running OpenMP on 128+ threads there is ~10x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def kernel_breakdown(model, img, qs, step_ms, reps=5):
    """Per-kernel-family time of one forward from HIP events recorded after every launch on the launch stream
    (cotr_set_profiling level 2).  An event costs ~3 us of stream time per launch, so the raw per-launch times are
    rescaled to sum to the un-instrumented step time measured above."""
    import collections
    model.set_profiling(2)
    import re
    fam = collections.OrderedDict()
    n_launch = 0
    for _ in range(reps):
        model(img, qs)
        torch.cuda.synchronize()
        prof = model.get_profile()
        n_launch = len(prof)
        for name, ms in prof:
            key = name.split(' ')[0]
            e = fam.setdefault(key, [0, 0.0, 0.0])
            e[0] += 1
            e[1] += ms
            mnk = re.search(r'(\d+)x(\d+)x(\d+)', name)          # GEMM / conv launches carry their M x N x K
            if mnk:
                e[2] += 2.0 * int(mnk.group(1)) * int(mnk.group(2)) * int(mnk.group(3))
    model.set_profiling(0)
    raw_total = sum(v[1] for v in fam.values()) / reps
    overhead = max(0.0, (raw_total - step_ms) / max(1, n_launch))      # ms per launch added by the event
    out = {}
    for k, (cnt, ms, fl) in fam.items():
        launches = cnt // reps
        us = max(0.0, ms / reps - overhead * launches) * 1e3
        out[k] = {'launches': launches, 'us': round(us, 1)}
        if fl > 0 and us > 0:
            out[k]['gflop'] = round(fl / reps / 1e9, 3)
            out[k]['tflops'] = round(fl / reps / us / 1e6, 1)
    dom = max(out, key=lambda k: out[k]['us'])
    res = {'launches_per_forward': n_launch, 'event_overhead_us_per_launch': round(overhead * 1e3, 2), 'families': out}
    if 'tflops' in out[dom]:
        res['dominant_family'] = {'name': dom, **out[dom], 'avg_us_per_launch': round(out[dom]['us'] / out[dom]['launches'], 2),
                                  'frac_of_fp32_mfma_peak': round(out[dom]['tflops'] / PEAK_FP32_MFMA_TFLOPS, 3)}
    return res


def other_regimes(sd, dev):
    """NOT the headline: the same path (a) with three independent calls in flight (three handles, three streams - the
    one-pair forward leaves CUs idle between its ~120 dependent launches) and (b) at the batched shapes the callers use."""
    import cotr_amd
    from cotr_amd.models import build_model
    from cotr_amd.utils.synth import synth_inputs
    out = {}
    models = []
    for _ in range(3):
        m = build_model(cotr_amd.default_args()).to(dev).eval()
        m.load_state_dict(sd)
        models.append(m)
    streams = [torch.cuda.Stream(device=dev) for _ in models]
    img, qs = synth_inputs(PAIRS_PER_GPU, QUERIES, seed=1)
    img, qs = img.to(dev), qs.to(dev)

    def run(n):
        for i in range(n):
            with torch.cuda.stream(streams[i % 3]):
                models[i % 3](img, qs)
    run(30)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(150)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 150
    out['three_calls_in_flight'] = {'ms_per_call': dt * 1e3, 'query_corr_per_s': QUERIES / dt,
                                    'tflops': flop(1, QUERIES) / dt / 1e12}
    m = models[0]
    for tag, b, q, n in (('batch_32_pairs_x_1000_queries', 32, 1000, 5), ('engine_batch_32_pairs_x_1_query', 32, 1, 10)):
        img, qs = synth_inputs(b, q, seed=2)
        img, qs = img.to(dev), qs.to(dev)
        for _ in range(2):
            m(img, qs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            m(img, qs)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        out[tag] = {'ms_per_call': dt * 1e3, 'query_corr_per_s': b * q / dt, 'pairs_per_s': b / dt,
                    'tflops': flop(b, q) / dt / 1e12, 'frac_of_fp32_mfma_peak': flop(b, q) / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS}
    return out


def hbm_traffic_bytes():
    """HBM<->L2 bytes per forward from the committed rocprofv3 PMC passes (profiles/r1_pmc_hbm_traffic.txt:
    FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate passes), or None."""
    path = os.path.join(ROOT, 'profiles', 'r1_pmc_hbm_traffic.txt')
    try:
        tot = 0.0
        for line in open(path):
            if line.startswith(('FETCH_SIZE', 'WRITE_SIZE')):
                tot += float(line.split('->')[1].split('MB')[0]) * 1e6
        return tot or None
    except (OSError, ValueError, IndexError):
        return None


def cpu_baseline(budget_s=12.0):
    from cotr_amd.utils.synth import synth_state_dict, synth_inputs
    from oracle import cotr_oracle
    cores = usable_cores()
    torch.set_num_threads(cores)
    sd = synth_state_dict(0)
    img, qs = synth_inputs(PAIRS_PER_GPU, QUERIES, seed=1)
    cotr_oracle.cotr_forward(sd, img, qs)  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        cotr_oracle.cotr_forward(sd, img, qs)
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or n >= 100:
            break
    return {'value': PAIRS_PER_GPU * QUERIES * n / dt, 'unit': 'query-correspondences/s', 'cores': cores,
            'kind': 'port',
            'sample': f'{n} forward calls of the same workload (1 pair x {QUERIES} queries, fp32) by oracle/cotr_oracle.py '
                      f'(torch CPU, {cores} threads), {dt:.1f} s'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='only the timed steps (for rocprofv3 runs)')
    ap.add_argument('--xcd-mapping', type=int, default=None, help='cotr_set_xcd_mapping policy (experiments)')
    ap.add_argument('--fused-stem', type=int, default=None, help='cotr_set_fused_stem (experiments)')
    ap.add_argument('--ffn-tail', type=int, default=None, help='cotr_set_ffn_tail (experiments)')
    ap.add_argument('--ffn-preln', type=int, default=None, help='cotr_set_ffn_preln (experiments)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if args.gpus > 1 and world == 1:
        raise SystemExit('for --gpus N > 1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs the MI355X (no CPU fallback of the product path)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', device_id=dev)

    import cotr_amd
    from cotr_amd.models import build_model
    from cotr_amd.utils.synth import synth_state_dict, synth_inputs

    if args.xcd_mapping is not None:
        from cotr_amd import _lib
        _lib.load_library().cotr_set_xcd_mapping(args.xcd_mapping)
    if args.ffn_preln is not None:
        from cotr_amd import _lib
        _lib.load_library().cotr_set_ffn_preln(args.ffn_preln)
    if args.ffn_tail is not None:
        from cotr_amd import _lib
        _lib.load_library().cotr_set_ffn_tail(args.ffn_tail)
    if args.fused_stem is not None:
        from cotr_amd import _lib
        _lib.load_library().cotr_set_fused_stem(args.fused_stem)
    model = build_model(cotr_amd.default_args()).to(dev).eval()
    model.load_state_dict(synth_state_dict(0))
    img, qs = synth_inputs(PAIRS_PER_GPU, QUERIES, seed=1 + rank)
    img, qs = img.to(dev), qs.to(dev)
    gathered = torch.empty((world * PAIRS_PER_GPU, QUERIES, 2), device=dev) if world > 1 else None

    def step():
        out = model(img, qs)['pred_corrs']
        if world > 1:  # RCCL all-gather of the predicted (x,y), asynchronous w.r.t. the next step's kernels
            return dist.all_gather_into_tensor(gathered, out, async_op=True)
        return None

    for _ in range(args.warmup):
        w = step()
        if w is not None:
            w.wait()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    works = []
    for i in range(args.steps):
        works.append(step())
        ev[i + 1].record()
    for w in works:
        if w is not None:
            w.wait()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = sum(ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)) / args.steps  # HIP events, launch stream

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        units = world * PAIRS_PER_GPU * QUERIES * args.steps
        achieved = flop(PAIRS_PER_GPU, QUERIES) / (kernel_ms * 1e-3) / 1e12
        line = {
            'metric': 'query-correspondences/sec at 256x512 SBS, 1k queries',
            'value': units / elapsed,
            'unit': 'query-correspondences/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': ms_per_step,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': 'BASELINE.json configs[1]: 1 pair 256x512 side-by-side x 1000 queries per GPU per step, '
                                   'zoom disabled, model(img[1,3,256,512], q[1,1000,2]); seeded random COTR weights',
                       'pairs_per_gpu': PAIRS_PER_GPU, 'queries_per_pair': QUERIES,
                       'parallelism': f'pairs sharded x{world}, all-gather of pred_corrs' if world > 1 else 'single GPU'},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved / PEAK_FP32_MFMA_TFLOPS, 'traffic': None,
                         'launch': 'one cotr_forward (all kernels of a step)', 'launch_ms_hip_events': kernel_ms,
                         'algorithmic_gflop_per_launch': flop(PAIRS_PER_GPU, QUERIES) / 1e9,
                         'min_hbm_gbs': min_hbm_bytes(PAIRS_PER_GPU, QUERIES) / (kernel_ms * 1e-3) / 1e9,
                         'hbm_peak_gbs': HBM_PEAK_GBS},
        }
        line['roofline']['traffic'] = hbm_traffic_bytes()
        line['roofline']['traffic_note'] = ('bytes per forward crossing L2<->fabric (MALL/HBM), rocprofv3 PMC FETCH_SIZE x2 + '
                                            'WRITE_SIZE from profiles/r1_pmc_hbm_traffic.txt; minimum is 75.4 MB')
        if world == 1 and not args.no_extras:
            line['roofline']['kernels'] = kernel_breakdown(model, img, qs, kernel_ms)
            line['also_measured'] = other_regimes(synth_state_dict(0), dev)
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

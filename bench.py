#!/usr/bin/env python
"""Headline benchmark: query-correspondences/sec of the COTR forward path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload headline|batch256|train]

Workloads
  headline (default)  BASELINE.json configs[1], SURVEY.md 8d "primary": one 256x512 side-by-side pair, 1000 queries, zoom
                      disabled -> one ``model(img[1,3,256,512], q[1,1000,2])`` call per step PER GPU (weak scaling).
  batch256            BASELINE.json configs[3]: 256 pairs x 1000 queries per step for the whole job, pairs sharded over the
                      ranks (32 per GPU on 8 GPUs, 256 on one), every rank holds only its own pairs (strong scaling).
  train               BASELINE.json configs[4] / SURVEY.md 8f4: one STAGE-2 ``COTRTrainer.train_batch`` step (cycle + bidirectional,
                      Adam, dropout 0.1, lr_backbone 1e-5: layer2 / layer3 of the backbone train) at 16 pairs x 200 queries per
                      GPU; value = pairs/s; ``--stage 1`` = the frozen-backbone first stage.  Not the headline metric.  The step is
                      ``cotr_amd.training.train_batch`` with the gradients in one flat buffer finished by one reduction launch
                      (GradSink) and Adam as one launch (FusedAdam); ``--no-grad-sink`` / ``--torch-adam`` = per-weight reductions /
                      torch's own Adam step, ``--graphed-train`` = the whole step as one captured HIP graph.
Synthetic data and seeded random weights of the COTR architecture (no checkpoint exists offline).  Inputs are resident in
HBM before the timed region.  N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): the predicted (x,y) of
every step are all-gathered over xGMI on RCCL's stream, overlapped with the next step; no collective in the math.

``python bench.py --gpus N`` with N > 1 and no torch.distributed environment starts its own ranks (it re-executes itself under
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1``); under an external launcher it reads
RANK / LOCAL_RANK / WORLD_SIZE as given.  ``--dry-run --backend gloo`` runs the same launch / timing / gather plumbing on the CPU
with a stand-in model (tests/test_bench_cpu.py): the only use of a non-HIP model, never a measurement.

Prints ONE JSON line on rank 0.  ``roofline`` is for the whole forward launch sequence (one "launch" = one cotr_forward =
the ~100 kernels of one step): achieved = FLOP(B,Q) / mean step time measured with HIP events on the launch stream, against
the fp32 MFMA peak (parity forces fp32 operands).  ``roofline.traffic`` = bytes crossing L2 <-> fabric per forward, measured
by this run itself (two rocprofv3 --pmc passes of a child process: FETCH_SIZE x2 on gfx950 + WRITE_SIZE) when rocprofv3 is
on the box, else taken from the committed profile - ``traffic_source`` says which.  ``cpu_baseline`` is the CPU oracle (a
torch-CPU restatement of the reference, kind "port"; the reference itself, kind "reference", where /root/reference is
importable) timed on this box's host cores on a bounded number of the same forward calls - rank 0, N == 1 only.
"""
import argparse
import csv
import glob
import json
import math
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

QUERIES = 1000
DENSE_QUERIES = 256 * 512     # --workload dense: the grid (j/512, i/256) of cotr_patch_flow_exhaustive.one_pass (inference_helper.py:116-127)
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, 2.4 GHz x 256 CUs
HBM_PEAK_GBS = 8000.0


KNOBS = []   # --set KNOB=INT pairs: applied to every model handle this run creates (tuning knobs are per handle, include/cotr_hip.h)


def apply_knobs(model):
    for name, val in KNOBS:
        model.set_knob(name, val)
    return model


def flop(b, q):
    """Algorithmic work of the path, SURVEY.md 2.2 / 8(d): per pair 24.641 GFLOP (backbone, input_proj,
    encoder, decoder K/V), per query 11.273 MFLOP (6 decoder layers + final norm + corr MLP once)."""
    return b * 24.641e9 + b * q * 11.273e6


# algorithmic work per stage of the launch sequence (cotr_set_profiling level 1 marks): per pair, the decoder per query
STAGE_GFLOP_PER_PAIR = {'stem+pool': 0.616563, 'layer1': 3.489661, 'layer2': 5.368709, 'layer3': 7.650410, 'input_proj': 0.268435,
                        'encoder': 6.442451, 'dec_kv': 0.805306}
DECODER_MFLOP_PER_QUERY = 11.273216


def stage_breakdown(model, img, qs, step_ms, reps=20):
    """roofline.stages: us, GFLOP, TFLOP/s and fraction of the fp32-MFMA peak per stage of one forward, from HIP events recorded
    on the launch stream at the 8 stage boundaries (cotr_set_profiling level 1: ~3 us of stream time per event, rescaled so that
    the stages sum to the un-instrumented step time)."""
    import collections
    b, q = qs.shape[0], qs.shape[1]
    model.set_profiling(1)
    acc = collections.OrderedDict()
    for _ in range(reps):
        model(img, qs)
        torch.cuda.synchronize()
        for name, ms in model.get_profile():
            acc[name] = acc.get(name, 0.0) + ms
    model.set_profiling(0)
    raw = sum(acc.values()) / reps
    scale = step_ms / raw if raw > 0 else 1.0
    out = collections.OrderedDict()
    for name, ms in acc.items():
        us = ms / reps * scale * 1e3
        gf = b * STAGE_GFLOP_PER_PAIR[name] if name in STAGE_GFLOP_PER_PAIR else (b * q * DECODER_MFLOP_PER_QUERY / 1e3 if name == 'decoder' else 0.0)
        e = {'us': round(us, 1), 'gflop': round(gf, 3)}
        if gf > 0 and us > 0:
            e['tflops'] = round(gf / us * 1e3, 1)
            e['frac'] = round(gf / us * 1e3 / PEAK_FP32_MFMA_TFLOPS, 3)
        out[name] = e
    return out


def dominant_kernel(n_forward=20, timeout_s=240):
    """roofline.dominant_kernel: the kernel symbol with the largest total time in a rocprofv3 --kernel-trace --stats pass of
    this script's own forward loop (child process), with its launches per forward and average duration."""
    exe = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if exe is None:
        return {'error': 'rocprofv3 not found'}
    tmp = tempfile.mkdtemp(prefix='cotr_kt_', dir='/tmp')
    try:
        cmd = [exe, '--kernel-trace', '--output-format', 'csv', '-d', tmp, '-o', 'kt', '--',
               sys.executable, os.path.abspath(__file__), '--traffic-child', str(n_forward)]
        r = subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           timeout=timeout_s)
        files = glob.glob(os.path.join(tmp, '**', '*kernel_trace.csv'), recursive=True)
        if r.returncode != 0 or not files:
            return {'error': f'rocprofv3 --kernel-trace failed (rc {r.returncode})'}
        rows = list(csv.DictReader(open(files[0])))
        rows.sort(key=lambda row: int(row['Start_Timestamp']))
        starts = [i for i, row in enumerate(rows) if 'stem_pool_kernel' in row['Kernel_Name']]
        if len(starts) < n_forward + 1:
            return {'error': f'expected {n_forward + 3} forwards in the trace, found {len(starts)}'}
        steady = rows[starts[-n_forward]:]
        tot, cnt = {}, {}
        for row in steady:
            k = row['Kernel_Name']
            tot[k] = tot.get(k, 0) + int(row['End_Timestamp']) - int(row['Start_Timestamp'])
            cnt[k] = cnt.get(k, 0) + 1
        busy = sum(tot.values())
        k = max(tot, key=tot.get)
        return {'name': k, 'launches_per_forward': cnt[k] / n_forward, 'avg_us': round(tot[k] / cnt[k] / 1e3, 2),
                'us_per_forward': round(tot[k] / n_forward / 1e3, 1), 'share_of_kernel_time': round(tot[k] / busy, 3),
                'kernel_busy_us_per_forward': round(busy / n_forward / 1e3, 1), 'launches_per_forward_all': len(steady) / n_forward,
                'source': f'rocprofv3 --kernel-trace child of this run, {n_forward} forwards (rocprofv3 adds ~2.5 us to every short kernel)'}
    except (subprocess.TimeoutExpired, OSError, ValueError, KeyError) as e:
        return {'error': f'{type(e).__name__}: {e}'}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


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
    import re
    model.set_profiling(2)
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
            for mnk in re.finditer(r'(\d+)x(\d+)x(\d+)', name):          # GEMM / conv launches carry their M x N x K
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


def time_calls(fn, n, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def other_regimes(sd, dev):
    """NOT the headline: the same path (a) with three independent calls in flight (three handles, three streams - the
    one-pair forward leaves CUs idle between its ~100 dependent launches) and (b) at the batched shapes the callers use."""
    import cotr_amd
    from cotr_amd.models import build_model
    from cotr_amd.utils.synth import synth_inputs
    out = {}
    models = []
    for _ in range(3):
        m = apply_knobs(build_model(cotr_amd.default_args()).to(dev).eval())
        m.load_state_dict(sd)
        models.append(m)
    streams = [torch.cuda.Stream(device=dev) for _ in models]
    img, qs = synth_inputs(1, QUERIES, seed=1)
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
    for tag, b, q, n in (('batch_32_pairs_x_1000_queries', 32, 1000, 5), ('engine_batch_32_pairs_x_1_query', 32, 1, 10),
                         ('dense_pass_1_pair_x_131072_queries', 1, 131072, 5)):
        img, qs = synth_inputs(b, q, seed=2)
        img, qs = img.to(dev), qs.to(dev)
        dt = time_calls(lambda: m(img, qs), n)
        out[tag] = {'ms_per_call': dt * 1e3, 'query_corr_per_s': b * q / dt, 'pairs_per_s': b / dt,
                    'tflops': flop(b, q) / dt / 1e12, 'frac_of_fp32_mfma_peak': flop(b, q) / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS}
    # the batch axis at the metric's query count (tools/frac_by_batch.py has the whole grid: profiles/r6_frac_by_batch.txt)
    by_batch = []
    for b in (1, 2, 4, 8, 16, 32):
        img, qs = synth_inputs(b, QUERIES, seed=2)
        img, qs = img.to(dev), qs.to(dev)
        dt = time_calls(lambda: m(img, qs), max(10, 120 // b), warm=3)
        by_batch.append({'pairs': b, 'queries': QUERIES, 'ms_per_call': round(dt * 1e3, 4),
                         'frac_of_fp32_mfma_peak': round(flop(b, QUERIES) / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)})
    out['by_batch'] = by_batch
    return out


def committed_by_batch(pairs, queries):
    """The measured point of profiles/r*_frac_by_batch.txt (1 GPU) nearest to (pairs, queries): what a rank of a multi-GPU run would run at."""
    import glob
    import re
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_frac_by_batch.txt'))):
        for ln in open(path):
            m = re.match(r'B=\s*(\d+) Q=\s*(\d+):\s*([\d.]+) ms\s+frac ([\d.]+)', ln)
            if m:
                b, q, ms, frac = int(m.group(1)), int(m.group(2)), float(m.group(3)), float(m.group(4))
                d = abs(math.log(b / max(pairs, 1))) + abs(math.log(q / max(queries, 1)))
                if best is None or d <= best[0]:
                    best = (d, {'pairs': b, 'queries': q, 'ms_per_call_1gpu': ms, 'frac_of_fp32_mfma_peak_1gpu': frac,
                                'source': os.path.relpath(path, ROOT)})
    return best[1] if best else None


# ---- research: split-f16 MFMAs (never the headline) ------------------------------------------------------------------
def research_split_f16_child():
    """Child process (COTR_HIP_EXPERIMENTAL=1: libcotr_hip_exp.so): the batched shapes with the RESEARCH knob split_f16=3 (fp32 products
    from three f16 MFMAs on packed split-f16 tensors, csrc/experimental/gemm_h2.h) next to the same library's fp32-MFMA path, and how far
    apart the two results are.  One JSON object on stdout."""
    import cotr_amd
    from cotr_amd.models import build_model
    from cotr_amd.utils.synth import synth_state_dict, synth_inputs
    dev = torch.device('cuda', 0)
    m = build_model(cotr_amd.default_args()).to(dev).eval()
    m.load_state_dict(synth_state_dict(0))
    out = {}
    for tag, b, q, n in (('batch_32_pairs_x_1000_queries', 32, 1000, 5), ('engine_batch_32_pairs_x_1_query', 32, 1, 10)):
        img, qs = synth_inputs(b, q, seed=2)
        img, qs = img.to(dev), qs.to(dev)
        res = {}
        for level in (0, 3):
            m.set_knob('split_f16', level)
            res[level] = m(img, qs)['pred_corrs'].clone()
            dt = time_calls(lambda: m(img, qs), n)
            key = 'fp32_mfma' if level == 0 else 'split_f16'
            out.setdefault(tag, {})[key] = {'ms_per_call': dt * 1e3, 'query_corr_per_s': b * q / dt, 'pairs_per_s': b / dt,
                                            'fp32_equivalent_tflops': flop(b, q) / dt / 1e12}
            if level:   # against ITS OWN roofline: 2517 TFLOP/s of dense f16 MFMA / 3 matrix instructions per fp32 product
                eq = flop(b, q) / dt / 1e12
                out[tag][key]['roofline'] = {'bound': 'mfma (f16, three instructions per fp32 product)', 'achieved': eq, 'peak': 839.0,
                                             'unit': 'TFLOP/s fp32-equivalent', 'frac': eq / 839.0}
        d = (res[0] - res[3]).abs() * torch.tensor([512.0, 256.0], device=dev)
        out[tag]['max_px_between_the_two_paths'] = float(d.max())
        out[tag]['speedup'] = out[tag]['fp32_mfma']['ms_per_call'] / out[tag]['split_f16']['ms_per_call']
    print(json.dumps(out), flush=True)


def research_split_f16():
    """also_measured['RESEARCH_split_f16_opt_in_not_the_product_path']: measured by a child on the experimental library; the headline, the
    roofline object and batched_frac stay on v_mfma_f32_32x32x2_f32 (dtype f32)."""
    note = ('RESEARCH, opt-in (knob split_f16 of libcotr_hip_exp.so, off by default, never on the product path): every fp32 product of the '
            'large GEMMs / convolutions / attention products (level 3) as three v_mfma_f32_32x32x16_f16 on packed split-f16 tensors (hi = f16(a), lo = f16((a - hi) * 2^11)); '
            'as close to the fp64 truth as the fp32-MFMA path on all 9 goldens of the reference (tests/test_experimental_gpu.py), not '
            'bit-identical to it; range: a pass that packs |x| >= 65504 anywhere is re-run on the fp32 kernels (cotr_h2_fallbacks); "fp32_equivalent_tflops" counts the fp32 work; its own roofline is 2517 / 3 = 839 '
            'TFLOP/s-equivalent (dense f16 MFMA peak over three matrix instructions per product): "roofline.frac" of each split_f16 entry')
    try:
        env = dict(os.environ, COTR_HIP_EXPERIMENTAL='1')
        p = subprocess.run([sys.executable, os.path.abspath(__file__), '--research-child'], env=env, capture_output=True, text=True, timeout=240)
        lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
        if p.returncode != 0 or not lines:
            return {'note': note, 'error': (p.stderr or p.stdout)[-300:]}
        return dict(json.loads(lines[-1]), note=note)
    except Exception as e:   # the research leg must never take the bench line down
        return {'note': note, 'error': repr(e)[:300]}


# ---- L2 <-> fabric traffic ------------------------------------------------------------------------------------------
def traffic_child(n_forward):
    """Child process of measure_traffic (runs under rocprofv3 --pmc): a few warm-up forwards, then n_forward forwards."""
    import cotr_amd
    from cotr_amd.models import build_model
    from cotr_amd.utils.synth import synth_state_dict, synth_inputs
    dev = torch.device('cuda', 0)
    model = apply_knobs(build_model(cotr_amd.default_args()).to(dev).eval())
    model.load_state_dict(synth_state_dict(0))
    img, qs = synth_inputs(1, QUERIES, seed=1)
    img, qs = img.to(dev), qs.to(dev)
    for _ in range(3 + n_forward):
        model(img, qs)
    torch.cuda.synchronize()


def measure_traffic(n_forward=5, timeout_s=240):
    """bytes per forward crossing L2 <-> fabric (Infinity Cache / HBM): FETCH_SIZE (KB; reports half the bytes of wide
    coalesced reads on gfx950: doubled, MI355X_MICROARCH.md) and WRITE_SIZE (KB) in SEPARATE rocprofv3 --pmc passes
    (they do not fit one pass), kernel trace only.  -> (bytes, per-counter dict) or (None, reason)."""
    exe = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if exe is None:
        return None, 'rocprofv3 not found'
    per = {}
    for ctr, corr in (('FETCH_SIZE', 2.0), ('WRITE_SIZE', 1.0)):
        tmp = tempfile.mkdtemp(prefix='cotr_pmc_', dir='/tmp')
        try:
            env = dict(os.environ, TMPDIR='/tmp')
            cmd = [exe, '--kernel-trace', '--pmc', ctr, '--output-format', 'csv', '-d', tmp, '-o', 'pmc', '--',
                   sys.executable, os.path.abspath(__file__), '--traffic-child', str(n_forward)]
            r = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s)
            files = glob.glob(os.path.join(tmp, '**', '*counter_collection.csv'), recursive=True)
            if r.returncode != 0 or not files:
                return None, f'rocprofv3 --pmc {ctr} failed (rc {r.returncode})'
            rows = [row for row in csv.DictReader(open(files[0])) if row.get('Counter_Name') == ctr]
            rows.sort(key=lambda row: int(row.get('Dispatch_Id', 0)))
            starts = [i for i, row in enumerate(rows) if 'stem_pool_kernel' in row['Kernel_Name']]   # first kernel of a forward
            if len(starts) < n_forward + 1:
                return None, f'{ctr}: expected {n_forward + 3} forwards in the trace, found {len(starts)}'
            steady = rows[starts[-n_forward]:]
            per[ctr] = sum(float(row['Counter_Value']) for row in steady) / n_forward * 1024.0 * corr
        except (subprocess.TimeoutExpired, OSError, ValueError, KeyError) as e:
            return None, f'{ctr}: {type(e).__name__}: {e}'
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return per['FETCH_SIZE'] + per['WRITE_SIZE'], per


def committed_traffic():
    """The same quantity from the newest committed rocprofv3 PMC summary under profiles/ (tools/pmc.sh), or None."""
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_hbm_traffic.txt'))):
        try:
            tot = 0.0
            for line in open(path):
                if line.startswith(('FETCH_SIZE', 'WRITE_SIZE')):
                    tot += float(line.split('->')[1].split('MB')[0]) * 1e6
            if tot:
                best = (tot, os.path.relpath(path, ROOT))
        except (OSError, ValueError, IndexError):
            pass
    return best


# ---- CPU baseline ----------------------------------------------------------------------------------------------------
def cpu_baseline(budget_s=12.0):
    from cotr_amd.utils.synth import synth_state_dict, synth_inputs
    from oracle import cotr_oracle, ref_import
    cores = usable_cores()
    torch.set_num_threads(cores)
    sd = synth_state_dict(0)
    img, qs = synth_inputs(1, QUERIES, seed=1)

    def timed(fn, budget):
        fn()  # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            fn()
            n += 1
            dt = time.perf_counter() - t0
            if dt > budget or n >= 1000:
                return n, dt
    n, dt = timed(lambda: cotr_oracle.cotr_forward(sd, img, qs, reference_cost=True), budget_s)
    out = {'value': QUERIES * n / dt, 'unit': 'query-correspondences/s', 'cores': cores, 'kind': 'port',
           'sample': f'{n} forward calls of the same workload (1 pair x {QUERIES} queries, fp32) by oracle/cotr_oracle.py '
                     f'(torch CPU, {cores} threads), {dt:.1f} s',
           'note': 'the port is timed doing everything the reference does, including what the reference computes and discards: '
                   'decoder.norm + corr_embed on all 6 decoder layers (cotr_model.py:37-39) and nn.MultiheadAttention\'s head-averaged '
                   'attention maps (reference_cost=True).  In the authoring container (8 cores) the port and the unmodified reference '
                   'time within a few percent of each other on these inputs (DESIGN.md section 6).  The reference (/root/reference, '
                   'Python) cannot travel to the GPU box: kind "port" there, a "reference" entry is added where it is importable'}
    if ref_import.reference_available():     # authoring container only: the unmodified reference, same inputs
        try:
            model = ref_import.build_reference_model()
            model.load_state_dict(sd)
            model.eval()
            with torch.no_grad():
                n2, dt2 = timed(lambda: model(img, qs), budget_s)
            out['reference'] = {'value': QUERIES * n2 / dt2, 'kind': 'reference', 'cores': cores,
                                'sample': f'{n2} calls of COTR.forward imported unchanged from /root/reference (torch CPU), {dt2:.1f} s'}
        except Exception as e:   # noqa: BLE001  (a baseline, never fatal)
            out['reference'] = {'error': f'{type(e).__name__}: {e}'}
    return out


# ---- training step -----------------------------------------------------------------------------------------------------
def run_train(args, dev, world, rank):
    import cotr_amd
    from cotr_amd import training
    from cotr_amd.models import build_model
    from cotr_amd.utils.synth import synth_state_dict
    pairs, nq = 16, 200                       # BASELINE.json configs[4]: bs=16 per GPU, 200 queries, cycle + bidirectional
    # stage 2 of the reference recipe (readme.md:50, train_cotr.py --lr_backbone=1e-5): layer2 / layer3 of the backbone train
    # (backbone.py:64-69) - what BASELINE.json configs[4] names; --stage 1 = the frozen-backbone first stage (lr_backbone 0)
    lr_backbone = 1e-5 if args.stage == 2 else 0.0
    model = apply_knobs(build_model(cotr_amd.default_args(dropout=0.1, lr_backbone=lr_backbone)).to(dev))
    model.load_state_dict(synth_state_dict(0))
    model.train()
    graphed = args.graphed_train
    use_sink = not args.no_grad_sink
    # Adam as one launch on the sink's flat buffers (training.FusedAdam: torch.optim.Adam's update and state dict)
    fused_adam = use_sink and not args.torch_adam
    optim = training.optimizer_for(model, learning_rate=1e-4, lr_backbone=lr_backbone, capturable=graphed, fused=fused_adam)
    g = torch.Generator().manual_seed(5 + rank)
    img = torch.randn(pairs, 3, 256, 512, generator=g).to(dev)
    query, target = torch.rand(pairs, nq, 2, generator=g).to(dev), torch.rand(pairs, nq, 2, generator=g).to(dev)
    # gradients in one flat buffer, finished by one reduction launch per backward pass (train_ops.GradSink; same values bit for bit)
    if graphed:      # the whole step (zero_grad .. optimizer step, gradient collectives included) as ONE captured HIP graph
        gstep = training.GraphedTrainStep(model, optim, img, query, target, warmup=max(args.warmup, 2), sink=use_sink)
        step = lambda: gstep(img, query, target)
    else:            # COTRTrainer.train_batch as the reference runs it: eager launches, loss.item() every step
        sink = training.grad_sink_for(optim) if use_sink else None
        step = lambda: training.train_batch(model, optim, img, query, target, sink=sink)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ident = rank_identity(world, dev)
    # FLOP of one step (DESIGN.md "Training step"): forward = backbone once (the cycle pass sees the same image) + input_proj, encoder and
    # decoder K/V projection on 2 x pairs (prediction and cycle pass, each with its own dropout) + two decodes; backward = 2 x the forward
    # FLOP of every trainable contraction (dX and dW): decoder, K/V projection, encoder, input_proj always, layer2 / layer3 in stage 2
    G = 1e9
    fwd = pairs * 17.126 * G + 2 * pairs * (0.268 + 6.442 + 0.805) * G + 2 * pairs * nq * 11.273e6
    bwd = 2 * (2 * pairs * (0.268 + 6.442 + 0.805) * G + 2 * pairs * nq * 11.273e6) + (2 * pairs * (5.37 + 7.65) * G if args.stage == 2 else 0.0)
    flop_step = fwd + bwd
    ms = elapsed / args.steps * 1e3
    if rank == 0:
        print(json.dumps({
            **ident,
            'roofline': {'bound': 'mfma', 'achieved': world * flop_step / ms / 1e9, 'peak': 157.3 * world, 'unit': 'TFLOP/s', 'frac': flop_step / ms / 1e9 / 157.3,
                         'flop_per_step_per_gpu': flop_step,
                         'flop_note': ('forward %.0f GFLOP (backbone once, input_proj / encoder / decoder K/V on 2 x %d pairs, two decodes of %d x %d queries) + backward %.0f GFLOP '
                                       '(dX and dW = 2 x the forward FLOP of every trainable contraction%s); traffic not measured for this workload'
                                       % (fwd / G, pairs, pairs, nq, bwd / G, ', layer2 / layer3 of the backbone included' if args.stage == 2 else '; backbone frozen')),
                         'traffic': None},
            'metric': 'training pairs/sec (COTRTrainer.train_batch step, cycle + bidirectional)', 'value': world * pairs * args.steps / elapsed,
            'unit': 'pairs/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': ('BASELINE.json configs[4]: stage-2 training step, 16 pairs x 200 queries per GPU, cycle consistency + '
                                    'bidirectional, dropout 0.1, Adam lr 1e-4, lr_backbone 1e-5 (layer2 / layer3 of the backbone train)'
                                    if args.stage == 2 else
                                    'stage 1 of the reference recipe (NOT configs[4]): the same step with the backbone frozen (lr_backbone 0)'),
                       'stage': args.stage, 'lr_backbone': lr_backbone,
                       'pairs_per_gpu': pairs, 'queries_per_pair': nq,
                       'step': ('captured HIP graph (GraphedTrainStep)' if graphed else
                                'eager train_batch (backward enqueued before the loss is read back; a NaN step is discarded afterwards)'),
                       'gradients': ('GradSink: flat buffer, one deferred reduction launch' + (' per third of the buffer, each third\'s reduce-scatter / all-gather started right behind it' if world > 1 and not graphed else '')) if use_sink else 'per-weight reductions + autograd accumulation',
                       'optimizer': 'FusedAdam (torch.optim.Adam update, one launch)' if fused_adam else 'torch.optim.Adam',
                       'parallelism': f'data parallel x{world}, reduce-scatter + all-gather of the gradients' if world > 1 else 'single GPU'},
        }), flush=True)


def reduce_elapsed(elapsed, steps, dev, world):
    """-> (max over the ranks of the timed region, [ms per step of every rank]); one all-gather of a scalar."""
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    every = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(every, t)
    vals = [float(v.item()) for v in every]
    return max(vals), [v / steps * 1e3 for v in vals]


def rank_identity(world, dev):
    """What a SCALE record can be checked against: the world size as the COMMUNICATOR sees it and the device every rank really
    runs on (gathered through the job's own process group; plain Python objects, outside the timed region)."""
    name = torch.cuda.get_device_name(dev) if dev.type == 'cuda' else 'cpu'
    mine = {'rank': dist.get_rank() if world > 1 else 0, 'device': str(dev), 'name': name,
            'pci_bus_id': (torch.cuda.get_device_properties(dev).pci_bus_id if dev.type == 'cuda' and
                           hasattr(torch.cuda.get_device_properties(dev), 'pci_bus_id') else None), 'pid': os.getpid()}
    if world <= 1:
        return {'rccl_ranks': 1, 'backend': None, 'ranks': [mine]}
    every = [None] * world
    dist.all_gather_object(every, mine)
    return {'rccl_ranks': dist.get_world_size(), 'backend': dist.get_backend(), 'ranks': every}


def dry_run(args, world, rank):
    """The launcher / rendezvous / barrier / max-over-ranks timing / gather / one-JSON-line plumbing of this file on the CPU
    (gloo), with a stand-in for the model: proves that `python bench.py --gpus N` starts its own ranks and that rank 0 alone
    prints the line, before a driver ever runs it on 8 GPUs.  Nothing here is a measurement of the product."""
    from cotr_amd.dist import all_gather_rows, shard_range
    if args.steps is None:
        args.steps = 3
    if args.warmup is None:
        args.warmup = 1
    dev = torch.device('cpu')
    if world > 1:
        dist.init_process_group('gloo')
    batch256, dense = args.workload == 'batch256', args.workload == 'dense'
    total_pairs = 8 if batch256 else (1 if dense else world)  # (a small stand-in for the 256 pairs)
    lo, hi = shard_range(total_pairs, world, rank)
    pairs = hi - lo
    counts = [shard_range(total_pairs, world, r)[1] - shard_range(total_pairs, world, r)[0] for r in range(world)]
    g = torch.Generator().manual_seed(1 if dense else 1 + rank)
    fake = lambda i, q: {'pred_corrs': q * 0.5 + i.mean(dim=(1, 2, 3)).view(-1, 1, 1)}
    finish = None
    if dense:
        # ONE pair, a stand-in grid of --dry-queries queries (default 1031: not a multiple of the world size; 131072 = the real
        # workload's grid), sharded by PairShardedModel exactly as the real workload: every rank holds the same inputs, decodes its slice, the all-gather is inside the step
        from cotr_amd.dist import PairShardedModel
        img, qs = torch.randn(1, 3, 16, 32, generator=g), torch.rand(1, args.dry_queries, 2, generator=g)
        calls = []

        def counted(i, q):
            calls.append(tuple(q.shape))
            return fake(i, q)
        sharded = PairShardedModel(counted)
        pairs = 1

        def step():
            out = sharded(img, qs)['pred_corrs']
            return (lambda: out[0]), None
    else:
        img, qs = torch.randn(pairs, 3, 16, 32, generator=g), torch.rand(pairs, 10, 2, generator=g)

        def step():
            out = fake(img, qs)['pred_corrs']
            return all_gather_rows(out, counts, async_op=True) if world > 1 else ((lambda: out), None)

    for _ in range(args.warmup):
        finish, w = step()
        if w is not None:
            w.wait()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    works = []
    for _ in range(args.steps):
        finish, w = step()
        works.append(w)
    for w in works:
        if w is not None:
            w.wait()
    gathered = finish()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    rank_ms = [elapsed / args.steps * 1e3]
    if world > 1:
        elapsed, rank_ms = reduce_elapsed(elapsed, args.steps, dev, world)
    ident = rank_identity(world, dev)
    if rank == 0:
        line = {'metric': 'DRY RUN (CPU stand-in model, gloo): plumbing only, not a measurement', 'value': total_pairs * 10 * args.steps / elapsed,
                'unit': 'query-correspondences/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                'ms_per_step': elapsed / args.steps * 1e3, 'ms_per_step_ranks': {'min': min(rank_ms), 'max': max(rank_ms)},
                'higher_is_better': True, 'scaling': 'strong' if (batch256 or dense) else 'weak', 'vs_baseline': None, 'dtype': 'f32',
                'data': 'synthetic', 'dry_run': True, 'gathered_rows': int(gathered.shape[0]),
                'config': {'workload': f'dry run of --workload {args.workload}', 'pairs_per_gpu': pairs}}
        line.update(ident)
        # what each rank of the REAL run would execute per step, and where that point sits on the measured 1-GPU batch curve
        real_pairs = {'headline': 1, 'batch256': -(-256 // world), 'dense': 1, 'train': 16}[args.workload]
        real_queries = -(-131072 // world) if dense else (200 if args.workload == 'train' else QUERIES)
        line['real_run_per_gpu'] = {'pairs': real_pairs, 'queries': real_queries, 'by_batch_point': committed_by_batch(real_pairs, real_queries)}
        if dense:
            line['dense_check'] = {'equals_unsharded': bool(torch.equal(gathered, fake(img, qs)['pred_corrs'][0])),
                                   'queries_this_rank': calls[-1][1], 'queries_total': int(qs.shape[1])}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=None)
    ap.add_argument('--workload', choices=['headline', 'batch256', 'dense', 'train'], default='headline',
                    help='headline: configs[1], one pair x 1000 queries per GPU; batch256: configs[3], 256 pairs sharded; dense: ONE pair x 131072 '
                         'grid queries (the dense initial pass, inference_helper.py:106-145) with the QUERIES sharded over the GPUs (strong scaling); '
                         'train: configs[4]')
    ap.add_argument('--torch-adam', action='store_true', help="--workload train: torch.optim.Adam's own multi-tensor step instead of FusedAdam")
    ap.add_argument('--no-grad-sink', action='store_true',
                    help='--workload train: per-weight gradient reductions + autograd accumulation instead of the GradSink')
    ap.add_argument('--graphed-train', action='store_true',
                    help='--workload train: the step as ONE captured HIP graph (training.GraphedTrainStep) instead of eager train_batch')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='only the timed steps (for rocprofv3 runs)')
    ap.add_argument('--traffic', choices=['auto', 'measure', 'committed', 'none'], default='auto',
                    help='roofline.traffic: measured by a rocprofv3 --pmc child of this run (auto: when available, N == 1, extras '
                         'on), or the committed profile')
    ap.add_argument('--traffic-child', type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument('--research', action='store_true',
                    help='also time the RESEARCH path (knob split_f16 of libcotr_hip_exp.so) in a child process and report it as a named secondary entry')
    ap.add_argument('--research-child', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--backend', choices=['nccl', 'gloo'], default='nccl', help='torch.distributed backend (nccl = RCCL; gloo with --dry-run)')
    ap.add_argument('--dry-queries', type=int, default=1031, help='--dry-run --workload dense: queries of the stand-in pair (131072 = the real grid)')
    ap.add_argument('--dry-run', action='store_true',
                    help='CPU plumbing check (tests): stand-in model, same launch / barrier / timing / gather / JSON code; never a measurement')
    ap.add_argument('--stage', type=int, choices=[1, 2], default=2,
                    help='--workload train: 2 (default, BASELINE.json configs[4]) = lr_backbone 1e-5, layer2/3 train; 1 = frozen backbone')
    ap.add_argument('--set', action='append', default=[], metavar='KNOB=INT',
                    help='experiments: cotr_set_knob(h, KNOB, INT) on every model handle (and the process-wide set) first, e.g. --set attention_fusion_max_rows=0')
    args = ap.parse_args()
    if args.traffic_child:
        return traffic_child(args.traffic_child)
    if args.research_child:
        return research_split_f16_child()
    if args.steps is None:
        args.steps = {'headline': 200, 'batch256': 5, 'dense': 10, 'train': 10}[args.workload]
    if args.warmup is None:
        args.warmup = {'headline': 20, 'batch256': 2, 'dense': 2, 'train': 3}[args.workload]

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` as the driver calls it: start one rank per GPU ourselves (torch.distributed.run on this node,
        # loopback rendezvous on a free port) and hand over; rank 0 of the child job prints the JSON line
        import socket
        with socket.socket() as sock:
            sock.bind(('127.0.0.1', 0))
            port = sock.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if args.dry_run:
        return dry_run(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs the MI355X (no CPU fallback of the product path)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group(args.backend, device_id=dev if args.backend == 'nccl' else None)

    import cotr_amd
    from cotr_amd import _lib
    from cotr_amd.dist import all_gather_rows, shard_range
    from cotr_amd.models import build_model
    from cotr_amd.utils.synth import synth_state_dict, synth_inputs

    global KNOBS
    KNOBS = [(kv.split('=')[0], int(kv.split('=')[1])) for kv in args.set]
    for name, val in KNOBS:                                   # the handle-less entry points (training ops) read the process-wide set
        _lib.set_knob(name, val)
    if args.workload == 'train':
        run_train(args, dev, world, rank)
        if world > 1:
            dist.destroy_process_group()
        return

    batch256, dense = args.workload == 'batch256', args.workload == 'dense'
    total_pairs = 256 if batch256 else (1 if dense else world)
    queries = DENSE_QUERIES if dense else QUERIES
    lo, hi = shard_range(total_pairs, world, rank)
    pairs = 1 if dense else hi - lo                           # this rank's pairs (batch256: its block of the 256; dense: THE pair)
    model = apply_knobs(build_model(cotr_amd.default_args()).to(dev).eval())
    model.load_state_dict(synth_state_dict(0))
    # dense: every rank holds the same pair and the same 256 x 512 query grid (inference_helper.py:116-127) and decodes ITS
    # slice of the queries (dist.PairShardedModel: pair x query shards; queries are independent, transformer.py:185-201)
    img, qs = synth_inputs(pairs, queries, seed=1 if dense else 1 + rank)
    if dense:
        jj, ii = torch.meshgrid(torch.arange(512), torch.arange(256), indexing='xy')
        qs = torch.stack([jj / 512.0, ii / 256.0], dim=-1).reshape(1, DENSE_QUERIES, 2).float()
    img, qs = img.to(dev), qs.to(dev)
    counts = [shard_range(total_pairs, world, r)[1] - shard_range(total_pairs, world, r)[0] for r in range(world)]
    my_queries = queries
    if dense and world > 1:
        from cotr_amd.dist import PairShardedModel
        sharded = PairShardedModel(model)
        q_lo, q_hi = shard_range(queries, world, rank)
        my_queries = q_hi - q_lo
    # one-time setup, not a step: pack the weights into the library (74 MB + the pos.W^T tables) and reserve its workspace, so
    # that a run with --warmup 0 does not time initialisation either.  No forward pass runs here.
    model.reserve(pairs, my_queries)
    torch.cuda.synchronize()

    def step():
        if dense and world > 1:                                 # encode the pair, decode this rank's query slice, all-gather: all timed
            sharded(img, qs)
            return None
        out = model(img, qs)['pred_corrs']
        if world > 1:  # RCCL all-gather of the predicted (x,y), asynchronous w.r.t. the next step's kernels
            return all_gather_rows(out, counts, async_op=True)[1]
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
    rank_ms = [elapsed / args.steps * 1e3]
    if world > 1:
        elapsed, rank_ms = reduce_elapsed(elapsed, args.steps, dev, world)
    kernel_ms = sum(ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)) / args.steps  # HIP events, launch stream
    ident = rank_identity(world, dev)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        units = total_pairs * queries * args.steps
        # this rank's algorithmic work per step (dense on several ranks: the whole encode + its slice of the queries)
        my_flop = 24.641e9 + my_queries * 11.273e6 if dense else flop(pairs, QUERIES)
        achieved = my_flop / (kernel_ms * 1e-3) / 1e12
        if dense:
            workload = ('the dense initial pass of cotr_flow (inference_helper.py:106-145): ONE pair 256x512 side-by-side x 131072 grid queries '
                        f'per step for the whole job; every rank encodes the pair and decodes {my_queries} of the queries '
                        '(dist.PairShardedModel), all-gather of pred_corrs inside the timed region; seeded random COTR weights')
        elif batch256:
            workload = ('BASELINE.json configs[3]: 256 pairs 256x512 side-by-side x 1000 queries per step for the whole job, '
                        f'{pairs} pairs on this GPU, model(img[{pairs},3,256,512], q[{pairs},1000,2]); seeded random COTR weights')
        else:
            workload = ('BASELINE.json configs[1]: 1 pair 256x512 side-by-side x 1000 queries per GPU per step, zoom disabled, '
                        'model(img[1,3,256,512], q[1,1000,2]); seeded random COTR weights')
        line = {
            'metric': 'query-correspondences/sec at 256x512 SBS, 1k queries',
            'value': units / elapsed,
            'unit': 'query-correspondences/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': ms_per_step,
            'ms_per_step_ranks': {'min': min(rank_ms), 'max': max(rank_ms)},
            'higher_is_better': True,
            'scaling': 'strong' if (batch256 or dense) else 'weak',
            'vs_baseline': None,
            'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': workload, 'pairs_per_gpu': pairs, 'queries_per_pair': queries,
                       'parallelism': (f'queries of one pair sharded x{world} (encode replicated), all-gather of pred_corrs' if dense and world > 1 else
                                       f'pairs sharded x{world}, all-gather of pred_corrs' if world > 1 else 'single GPU')},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved / PEAK_FP32_MFMA_TFLOPS, 'traffic': None,
                         'launch': 'one cotr_forward (all kernels of a step)', 'launch_ms_hip_events': kernel_ms,
                         'algorithmic_gflop_per_launch': my_flop / 1e9,
                         'min_hbm_gbs': min_hbm_bytes(pairs, my_queries) / (kernel_ms * 1e-3) / 1e9,
                         'hbm_peak_gbs': HBM_PEAK_GBS,
                         'peak_at_measured_clock': {'one_pair_2.38GHz': 155.9, 'all_cus_dense_mfma_2.15-2.25GHz': [140.9, 147.5]},
                         'clock_note': ('the yardstick is the nominal 157.3 TFLOP/s (65536 FLOP/clock x 2.4 GHz).  What the chip clocks to depends on the load: an '
                                        'un-instrumented probe wavefront (s_memtime vs the 100 MHz wall clock, tools/clock_settle.py) reads 2.38-2.39 GHz during '
                                        'the 1-pair forward and during kernels whose matrix pipes are 58-66 % busy (profiles/r4_shader_clock_probe_and_smi.txt: '
                                        '155.9 TFLOP/s); with all 256 CUs on dense matrix work (ffn_rows / att_rows, matrix pipes ~0.8 busy) the phase stamps of '
                                        'round 5 read 2.15-2.25 GHz (docs/LABNOTES.md, round 5 lab record; profiles/r5_final_ffn_rows_probe.txt): the power-limited '
                                        'ceiling of the batched regime is 0.90-0.94 of the nominal peak, so batched_frac 0.72 is ~0.78 of what the chip can clock')},
        }
        research = [f'{n}={v}' for n, v in KNOBS if n.startswith('split_f16') and v]
        if research:   # --set split_f16=N on the experimental library: the line is NOT a measurement of the fp32-MFMA product path - say so in it
            line['dtype'] = 'f32 carried as packed split-f16 (RESEARCH: three f16 MFMAs per fp32 product, docs/LABNOTES.md 3e)'
            line['RESEARCH_not_the_product_path'] = ('measured with --set ' + ' '.join(research) + ' on libcotr_hip_exp.so: results are as close to fp64 as the '
                                                     'fp32-MFMA path but not bit-identical to it; "roofline" below divides fp32-EQUIVALENT work by the fp32-MFMA '
                                                     'peak and is not a fraction of any roofline of the kernels that ran')
            line['roofline']['bound'] = 'none (research path)'
        extras = world == 1 and not args.no_extras and not batch256 and not dense and not research
        line.update(ident)
        roof = line['roofline']
        mode = args.traffic
        if mode == 'auto':
            mode = 'measure' if extras else 'committed'
        if mode == 'measure':
            tot, detail = measure_traffic()
            if tot is not None:
                roof['traffic'], roof['traffic_source'] = tot, 'measured by this run: rocprofv3 --pmc child, 5 forwards'
                roof['traffic_detail'] = {k: round(v) for k, v in detail.items()}
            else:
                roof['traffic_measure_error'] = detail
                mode = 'committed'
        if mode == 'committed' and not batch256 and not dense:
            c = committed_traffic()
            if c is not None:
                roof['traffic'], roof['traffic_source'] = c[0], f'committed profile {c[1]}'
        roof['traffic_note'] = ('bytes per forward crossing L2<->fabric (Infinity Cache / HBM): rocprofv3 PMC FETCH_SIZE x2 (gfx950) + '
                                f'WRITE_SIZE, separate passes; algorithmic minimum {min_hbm_bytes(pairs, QUERIES) / 1e6:.1f} MB')
        if extras:
            roof['stages'] = stage_breakdown(model, img, qs, kernel_ms)
            roof['dominant_kernel'] = dominant_kernel()
            roof['kernels'] = kernel_breakdown(model, img, qs, kernel_ms)
            roof['launches_per_forward'] = roof['kernels']['launches_per_forward']
            line['also_measured'] = other_regimes(synth_state_dict(0), dev)
            roof['batched_frac'] = line['also_measured']['batch_32_pairs_x_1000_queries']['frac_of_fp32_mfma_peak']
            roof['batched_frac_note'] = 'same path at 32 pairs x 1000 queries per call (throughput regime)'
            roof['by_batch'] = line['also_measured'].pop('by_batch')
            if args.research:   # opt-in: the driver's run times the product only
                line['also_measured']['RESEARCH_split_f16_opt_in_not_the_product_path'] = research_split_f16()
        if world == 1 and not args.no_cpu_baseline and not batch256 and not dense:
            line['cpu_baseline'] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

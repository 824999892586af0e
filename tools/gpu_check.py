"""One-shot diagnostic for a GPU box: runs every kernel-level and stage-level comparison WITHOUT
stopping at the first failure and prints a table (also written to gpurun_out/gpu_check.json).
Usage:  python tools/gpu_check.py
Test infrastructure (imports oracle/)."""
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

results = {}


def record(name, fn):
    t = time.time()
    try:
        results[name] = fn()
    except Exception as e:  # noqa: BLE001
        results[name] = f'EXC {type(e).__name__}: {e}'
        traceback.print_exc()
    print(f'{name:40s} {results[name]}   [{time.time() - t:.1f}s]', flush=True)


def main():
    from tests import gpu_helpers as G
    from tests import test_ops_gpu as T
    import cotr_amd
    from cotr_amd.models import build_model
    from cotr_amd.utils.synth import synth_state_dict, synth_inputs
    from oracle import cotr_oracle

    print('device', torch.cuda.get_device_name(0), 'count', torch.cuda.device_count(), flush=True)

    def run_test(fn, *a):
        def go():
            fn(*a)
            torch.cuda.synchronize()
            return 'ok'
        return go

    record('op.lin_sine', run_test(T.test_lin_sine_encoding))
    record('op.layernorm', run_test(T.test_layernorm))
    for shp in [(1, 64, 32), (1000, 256, 256), (77, 256, 1024), (24576, 128, 64), (65536, 128, 64)]:
        record(f'op.linear{shp}', run_test(T.test_linear_bias_relu_residual, *shp))
    record('op.linear_pos_bn', run_test(T.test_linear_pos_prologue_and_bn_epilogue))
    for c in [(1, 16, 64, 64, 3, 1), (2, 16, 128, 128, 3, 2), (1, 32, 256, 512, 1, 2), (1, 64, 64, 256, 1, 1)]:
        record(f'op.conv{c}', run_test(T.test_conv_frozenbn_residual_relu, *c))
    record('op.stem+pool', run_test(T.test_stem_conv7x7_and_maxpool))
    for a in [(2, 200, 1.0), (1, 512, 1.0), (3, 1, 1.0), (1, 129, 6.0)]:
        record(f'op.attention{a}', run_test(T.test_attention, *a))

    sd = synth_state_dict(0)
    m = build_model(cotr_amd.default_args()).cuda().eval()
    m.load_state_dict(sd)
    img, qs = synth_inputs(2, 257, seed=2)
    taps = {}
    ref = cotr_oracle.cotr_forward(sd, img, qs, taps=taps)
    out = [None]

    m.set_debug_taps(True)

    def fwd():
        out[0] = m(img.cuda(), qs.cuda())['pred_corrs'].cpu()
        return 'ok'
    record('model.forward(2,257)', fwd)
    checks = {
        'stem': G.nchw_to_sbs(taps['stem']), 'pool': G.nchw_to_sbs(taps['pool']),
        'layer1': G.nchw_to_sbs(taps['layer1.2']), 'layer2': G.nchw_to_sbs(taps['layer2.3']),
        'layer3': G.nchw_to_sbs(taps['layer3.5']), 'src': G.seq_to_rows(taps['src']),
        'pos': taps['pos'][:, 0], 'memory': G.seq_to_rows(taps['enc.5']),
        'query_pos': G.seq_to_rows(taps['query_pos']),
    }
    for k, v in checks.items():
        record('tap.' + k, lambda k=k, v=v: f'rel_err {G.rel_err(m.debug_tap(k).cpu().view(v.shape), v):.3e}')
    m.set_debug_taps(False)
    if out[0] is not None:
        record('model.px_err(2,257)', lambda: f'{cotr_oracle.px_err(out[0], ref):.3e} px')
    img1, q1 = synth_inputs(1, 1000, seed=1)
    ref1 = cotr_oracle.cotr_forward(sd, img1, q1)
    ref1_64 = cotr_oracle.cotr_forward(sd, img1, q1, dtype=torch.float64)
    record('model.px_err(1,1000) vs f32', lambda: f'{cotr_oracle.px_err(m(img1.cuda(), q1.cuda())["pred_corrs"].cpu(), ref1):.3e} px')
    record('model.px_err(1,1000) vs f64', lambda: f'{cotr_oracle.px_err(m(img1.cuda(), q1.cuda())["pred_corrs"].cpu(), ref1_64):.3e} px')

    def timing():
        i, q = img1.cuda(), q1.cuda()
        for _ in range(50):   # the GPU idles (low clocks) while the CPU oracle runs: warm it up
            m(i, q)
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(20):
            m(i, q)
        torch.cuda.synchronize()
        return f'{(time.time() - t) / 20 * 1e3:.3f} ms / forward (B=1,Q=1000)'
    record('model.time', timing)

    def prof():
        m.set_profiling(True)
        m(img1.cuda(), q1.cuda())
        torch.cuda.synchronize()
        p = m.get_profile()
        m.set_profiling(False)
        return ' '.join(f'{n}={t:.3f}' for n, t in p)
    record('model.stage_ms', prof)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'gpu_check.json'), 'w') as f:
        json.dump(results, f, indent=1)


if __name__ == '__main__':
    main()

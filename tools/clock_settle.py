"""Which shader clock do the production kernels really run at?  (round-3 verdict, weak 7: GRBM_GUI_ACTIVE / time read 2.4-4.5 "GHz",
an instrumented GEMM read 2.04-2.2 GHz from s_memtime - one of the two is wrong.)

A one-wavefront probe kernel (tools/micro/clock_probe.hip) runs on its OWN stream beside the un-instrumented library kernels and
samples s_memtime (shader cycles) against s_memrealtime (100 MHz) in 100 us windows; in parallel a host thread polls the
driver's sclk (sysfs pp_dpm_sclk / rocm-smi / amd-smi).  Workloads, ~2 s each: idle, the 1-pair forward, the 32-pair forward,
a K-deep large-tile GEMM back to back, a K = 256 GEMM back to back.   GPU box:  python tools/clock_settle.py
"""
import ctypes, glob, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cotr_amd
from cotr_amd import _lib
from cotr_amd.models import build_model
from cotr_amd.utils.synth import synth_state_dict, synth_inputs

HERE = os.path.dirname(os.path.abspath(__file__))
probe = ctypes.CDLL(os.path.join(HERE, 'micro', 'libclock_probe.so'))
probe.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
lib = _lib.load_library()
dev = torch.device('cuda:0')
WINDOW = 10000            # 100 us of the 100 MHz wall clock
SAMPLES = 30000           # hard bound: 3 s of probe, whatever happens to the stop flag


def sysfs_sclk():
    out = []
    for f in glob.glob('/sys/class/drm/card*/device/pp_dpm_sclk'):
        try:
            for line in open(f):
                if '*' in line:
                    out.append(line.strip())
        except OSError:
            pass
    return out


class Poller(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.stop = False
        self.sysfs, self.smi = [], []

    def run(self):
        t0 = time.time()
        did_smi = False
        while not self.stop:
            s = sysfs_sclk()
            if s:
                self.sysfs.append(s[0])
            if not did_smi and time.time() - t0 > 0.8:       # one smi reading in the middle of the workload
                did_smi = True
                for cmd in (['rocm-smi', '--showclocks'], ['amd-smi', 'metric', '--clock']):
                    try:
                        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=20)
                        keep = [l.strip() for l in r.stdout.splitlines() if any(k in l.lower() for k in ('sclk', 'gfx', 'clk'))]
                        self.smi.append((cmd[0], keep[:14]))
                    except Exception as e:                    # noqa: BLE001
                        self.smi.append((cmd[0], [repr(e)]))
            time.sleep(0.05)


def measure(tag, body, seconds=2.0, flop_per_call=None):
    side = torch.cuda.Stream()
    out = torch.zeros((SAMPLES, 3), dtype=torch.int64, device=dev)
    stop = torch.zeros(1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    poll = Poller()
    poll.start()
    probe.clock_probe_launch(out.data_ptr(), SAMPLES, WINDOW, stop.data_ptr(), side.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time()
    n = 0
    e0.record()
    while time.time() - t0 < seconds:
        body()
        n += 1
        if n % 8 == 0:
            torch.cuda.current_stream().synchronize()
    e1.record()
    torch.cuda.current_stream().synchronize()
    ms = e0.elapsed_time(e1)
    with torch.cuda.stream(torch.cuda.Stream()):
        stop.fill_(1)
    side.synchronize()
    poll.stop = True
    poll.join()
    v = out.cpu().numpy()
    v = v[v[:, 1] > 0]
    # samples taken while the workload ran (drop the first / last 5 %)
    k = len(v)
    v = v[k // 20: k - k // 20]
    ghz = v[:, 0] / v[:, 1] * 0.1
    line = f'{tag:44s} probe: {len(v):5d} windows  shader clock median {np.median(ghz):.3f} GHz  p5 {np.percentile(ghz, 5):.3f}  p95 {np.percentile(ghz, 95):.3f}'
    if flop_per_call:
        tf = flop_per_call * n / ms / 1e9
        peak = 1024 * 64 * np.median(ghz) / 1e3            # 1024 SIMDs x 64 FLOP / clock
        line += f' | {ms / n:8.3f} ms/call  {tf:6.1f} TFLOP/s = {tf / 157.3:.3f} of 157.3, {tf / peak:.3f} of the {peak:.1f} TFLOP/s this clock gives'
    print(line)
    from collections import Counter
    if poll.sysfs:
        print('    sysfs pp_dpm_sclk (current level, polls):', dict(Counter(poll.sysfs)))
    for name, lines in poll.smi:
        print(f'    {name}:', ' | '.join(lines))
    sys.stdout.flush()
    return float(np.median(ghz))


m = build_model(cotr_amd.default_args()).cuda().eval()
m.load_state_dict(synth_state_dict(0))
P = lambda t: t.data_ptr()
sp = _lib.current_stream_ptr()

measure('idle (probe alone)', lambda: time.sleep(0.01), seconds=1.0)
for B, Q in ((1, 1000), (32, 1000)):
    img, qs = synth_inputs(B, Q, seed=1)
    img, qs = img.cuda(), qs.cuda()
    for _ in range(3):
        m(img, qs)
    measure(f'forward {B} pair(s) x {Q} queries', lambda: m(img, qs), flop_per_call=B * 24.641e9 + B * Q * 11.273e6)
for tag, M, N, K, cfg in (('large tile GEMM 16384x1024x1024 (cfg 26)', 16384, 1024, 1024, 26), ('large tile GEMM 16384x256x1024 (cfg 27)', 16384, 256, 1024, 27),
                          ('large tile GEMM 262144x256x256 (cfg 27)', 262144, 256, 256, 27), ('4096^3 (cfg 26)', 4096, 4096, 4096, 26)):
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    y = torch.empty(M, N, device=dev)

    def body():
        for _ in range(10):
            lib.cotr_op_linear_cfg(P(x), P(w), None, None, 0, P(y), M, N, K, cfg, sp)
    body()
    measure(tag, body, flop_per_call=10 * 2.0 * M * N * K)

"""Throughput with N independent forwards in flight (N model handles on N streams) vs one. GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cotr_amd
from cotr_amd.models import build_model
from cotr_amd.utils.synth import synth_state_dict, synth_inputs

sd = synth_state_dict(0)
for nstreams in (1, 2, 3, 4):
    models = []
    for i in range(nstreams):
        m = build_model(cotr_amd.default_args()).cuda().eval()
        m.load_state_dict(sd)
        models.append(m)
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    img, qs = synth_inputs(1, 1000, seed=1)
    img, qs = img.cuda(), qs.cuda()
    torch.cuda.synchronize()
    def run(n):
        for i in range(n):
            k = i % nstreams
            with torch.cuda.stream(streams[k]):
                models[k](img, qs)
    run(20 * nstreams)
    torch.cuda.synchronize()
    n = 200
    t = time.perf_counter()
    run(n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f'{nstreams} in flight: {dt / n * 1e3:.3f} ms per forward, {1000 * n / dt:.0f} query-corr/s', flush=True)
    del models

"""HBM bandwidth probes with torch ops (fill = pure write, copy = read + write, sum = pure read). GPU box."""
import torch, time
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for mb in (67, 268, 1072):
    n = mb * (1 << 20) // 4
    a = torch.empty(n, device='cuda'); b = torch.empty(n, device='cuda')
    a.normal_()
    us = t(lambda: a.fill_(1.0)); print(f'{mb:5d} MB fill  {us:8.1f} us  {mb * 1.048576e6 / us / 1e6:6.2f} TB/s write')
    us = t(lambda: b.copy_(a)); print(f'{mb:5d} MB copy  {us:8.1f} us  {2 * mb * 1.048576e6 / us / 1e6:6.2f} TB/s read+write')
    us = t(lambda: a.sum()); print(f'{mb:5d} MB sum   {us:8.1f} us  {mb * 1.048576e6 / us / 1e6:6.2f} TB/s read')
    us = t(lambda: torch.add(a, b, out=b)); print(f'{mb:5d} MB add   {us:8.1f} us  {3 * mb * 1.048576e6 / us / 1e6:6.2f} TB/s 2 reads + write')

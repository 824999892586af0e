"""Sum FETCH_SIZE / WRITE_SIZE (KB) over the library's kernels of the steady-state steps of a rocprofv3 --pmc run.
    python tools/pmc_summary.py gpurun_out/pmc_r1_FETCH_SIZE/r1_counter_collection.csv [steps=10] [warmup=3]
FETCH_SIZE on gfx950 reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM): doubled below."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ours = ('gemm_kernel', 'gemm_ks_kernel', 'gemm_big_kernel', 'gemm_wp_kernel', 'gemm_ws_kernel', 'gemm_wp_dual', 'bottleneck_kernel', 'stem_pool_kernel', 'attention_kernel', 'layernorm_kernel', 'posenc_kernel', 'maxpool_kernel',
        'head2_kernel', 'ffn_fused_kernel', 'ln_reduce_kernel', 'gemm_ks_dual_kernel', 'gemm_big_dual_kernel', 'attention_wide_kernel',
        'dec_head_kernel')
rows = [r for r in csv.DictReader(open(path)) if any(k in r['Kernel_Name'] for k in ours)]
n_fwd = sum(1 for r in rows if ('gemm_kernel<2, 2, 2, 1, 2>' in r['Kernel_Name'] or 'stem_pool_kernel' in r['Kernel_Name']))
per_step = len(rows) // n_fwd if n_fwd else len(rows) // (steps + warm)
rows = rows[: per_step * (steps + warm)]
steady = rows[warm * per_step:]
name = steady[0]['Counter_Name']
tot = sum(float(r['Counter_Value']) for r in steady) / steps
fam = defaultdict(float)
for r in steady:
    fam[r['Kernel_Name'].split('(')[0][:40]] += float(r['Counter_Value']) / steps
corr = 2.0 if name == 'FETCH_SIZE' else 1.0
print(f'{name}: {per_step} launches/step, {tot:.0f} KB/step raw, x{corr:.0f} correction -> {tot * corr * 1024 / 1e6:.1f} MB per forward')
for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:8]:
    print(f'   {k:42s} {v * corr * 1024 / 1e6:8.1f} MB')

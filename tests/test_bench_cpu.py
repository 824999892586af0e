"""bench.py's launch plumbing without a GPU: `python bench.py --gpus 2` must start its own two ranks (the driver calls it exactly
like that), rendezvous on loopback, time the steps between barriers, gather, and print ONE JSON line on rank 0."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env=None):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + argv, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout                              # rank 0 alone prints, exactly one line
    return json.loads(lines[0])


def _clean_env():
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    return env


def test_bench_starts_its_own_ranks():
    line = _run(['--gpus', '2', '--steps', '3', '--warmup', '1', '--backend', 'gloo', '--dry-run'], _clean_env())
    assert line['dry_run'] is True and line['n_gpus'] == 2 and line['steps'] == 3 and line['warmup'] == 1
    assert line['scaling'] == 'weak' and line['gathered_rows'] == 2          # one pair per rank, both gathered
    assert 0 < line['ms_per_step_ranks']['min'] <= line['ms_per_step_ranks']['max']
    assert abs(line['ms_per_step'] - line['ms_per_step_ranks']['max']) < 1e-9     # the slowest rank's clock is the job's


def test_bench_batch_workload_shards_pairs_and_gathers_inside_the_timed_region():
    line = _run(['--gpus', '2', '--workload', 'batch256', '--backend', 'gloo', '--dry-run'], _clean_env())
    assert line['n_gpus'] == 2 and line['scaling'] == 'strong'
    assert line['config']['pairs_per_gpu'] == 4 and line['gathered_rows'] == 8   # 8 stand-in pairs over 2 ranks


def test_bench_single_rank_dry_run():
    line = _run(['--dry-run', '--backend', 'gloo'], _clean_env())
    assert line['n_gpus'] == 1 and line['gathered_rows'] == 1


def test_bench_dense_workload_shards_the_queries_of_one_pair():
    """--workload dense: ONE pair x the query grid, the QUERIES sharded over the ranks through dist.PairShardedModel (each rank
    encodes the pair, decodes its slice), all-gather inside the timed region, strong scaling; the gathered prediction equals
    the unsharded one and the line says which ranks / devices the communicator really had."""
    line = _run(['--gpus', '2', '--workload', 'dense', '--backend', 'gloo', '--dry-run'], _clean_env())
    assert line['n_gpus'] == 2 and line['scaling'] == 'strong' and line['dry_run'] is True
    assert line['dense_check'] == {'equals_unsharded': True, 'queries_this_rank': 516, 'queries_total': 1031}
    assert line['gathered_rows'] == 1031
    assert line['rccl_ranks'] == 2 and line['backend'] == 'gloo'
    assert [r['rank'] for r in line['ranks']] == [0, 1] and len({r['pid'] for r in line['ranks']}) == 2
    assert all(r['name'] == 'cpu' for r in line['ranks'])


def test_every_multi_rank_line_names_its_ranks():
    line = _run(['--gpus', '2', '--steps', '2', '--warmup', '0', '--backend', 'gloo', '--dry-run'], _clean_env())
    assert line['rccl_ranks'] == 2 and len(line['ranks']) == 2
    single = _run(['--workload', 'dense', '--dry-run', '--backend', 'gloo'], _clean_env())
    assert single['rccl_ranks'] == 1 and single['dense_check']['queries_this_rank'] == 1031


def test_dense_workload_on_eight_ranks_shards_the_full_query_grid():
    """SURVEY 8(e) without the node: `python bench.py --workload dense --gpus 8` starts eight ranks, shards the 131 072 queries of the
    dense pass (inference_helper.py:105-165) 16 384 per rank, gathers them inside the timed region and names its eight ranks."""
    line = _run(['--gpus', '8', '--workload', 'dense', '--backend', 'gloo', '--dry-run', '--dry-queries', '131072', '--steps', '2', '--warmup', '1'],
                _clean_env())
    assert line['n_gpus'] == 8 and line['rccl_ranks'] == 8 and line['scaling'] == 'strong'
    assert line['dense_check'] == {'equals_unsharded': True, 'queries_this_rank': 16384, 'queries_total': 131072}
    assert line['gathered_rows'] == 131072
    assert [r['rank'] for r in line['ranks']] == list(range(8)) and len({r['pid'] for r in line['ranks']}) == 8

"""bench.py --gpus N without a launcher environment starts N ranks itself (torch.distributed.run as a child process) and relays
rank 0's single JSON line.  Run here on CPU through LAFF_BENCH_DRYRUN=1: gloo instead of RCCL, the oracle-backed stand-in
kernels of tests/dist_util.py on a toy problem -- the launch, rendezvous, collective and reporting path of an N > 1 run."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n, *extra):
    env = dict(os.environ, LAFF_BENCH_DRYRUN='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--steps', '2', '--warmup', '1'] + list(extra),
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout                     # ONE line on stdout, whatever the ranks printed elsewhere
    return json.loads(lines[0])


def test_self_launch_two_ranks_dryrun():
    line = _run(2)
    assert line['n_gpus'] == 2 and line['rccl_ranks'] == 2
    assert line['steps'] == 2 and line['warmup'] == 1 and line['scaling'] == 'strong'
    assert line['config']['shard'] == 'video'              # the decomposition BASELINE.json names is the headline
    assert line['alt_shard']['shard'] == 'text' and line['alt_shard']['ranks_equal']
    assert [a['shard'] for a in line['alt_shards']] == ['text', 'video16'] and all(a['ranks_equal'] for a in line['alt_shards'])
    b16, b32 = line['alt_shards'][1]['gathered_bytes_per_step'], line['gathered_bytes_per_step']
    assert b16['all_gather_text_16bit'] * 2 == b32['all_gather_text_fp32'] and b16['all_to_all_pairs'] > 0
    assert line['value'] > 0 and line['ms_per_step'] > 0


def test_self_launch_two_ranks_shard_auto_and_video_agree_dryrun():
    """`--shard auto` and `--shard video` through the self-launch (a child torch.distributed.run, never an exec): the default IS
    'video' (BASELINE.json's decomposition), 'auto' picks by the smaller side, every line names its scheme, reports the other two
    beside it with their gathered bytes, and all of them end with the same ranks and the same seven metrics."""
    from laff_amd.dist import choose_sharding
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from dist_util import problem
    xt, xv = problem()[:2]
    dflt, vid, auto = _run(2), _run(2, '--shard', 'video'), _run(2, '--shard', 'auto')
    assert dflt['config']['shard'] == 'video' and vid['config']['shard'] == 'video'
    assert auto['config']['shard'] == choose_sharding(len(xt), len(xv)) and auto['config']['shard_arg'] == 'auto'
    for line in (vid, auto):
        assert line['rccl_ranks'] == 2 and line['n_gpus'] == 2
        names = [line['config']['shard']] + [a['shard'] for a in line['alt_shards']]
        assert sorted(names) == ['text', 'video', 'video16']
        assert all(a['ranks_equal'] for a in line['alt_shards'])
        assert all(sum(a['gathered_bytes_per_step'].values()) > 0 for a in line['alt_shards'])
        assert sum(line['gathered_bytes_per_step'].values()) > 0
    assert vid['quality'] == auto['quality']
    # 'video' gathers the fp32 text rows, 'text' the fp32 video rows: the bytes follow the sides
    gv = vid['gathered_bytes_per_step']
    gt_ = (auto if auto['config']['shard'] == 'text' else vid)
    tb = gt_['gathered_bytes_per_step'] if gt_['config']['shard'] == 'text' else [a for a in vid['alt_shards'] if a['shard'] == 'text'][0]['gathered_bytes_per_step']
    assert gv['all_gather_text_fp32'] * len(xv) == tb['all_gather_video_fp32'] * len(xt)


def test_single_rank_needs_no_launcher_dryrun():
    line = _run(1)
    assert line['n_gpus'] == 1 and line['rccl_ranks'] == 1


def test_more_gpus_than_the_node_has_is_refused():
    """Without the dry-run switch the parent counts the visible GPUs before starting anything (0 in this container)."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip('this node has the GPUs')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'LAFF_BENCH_DRYRUN')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1'], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 2 and 'visible GPU' in out.stderr and out.stdout.strip() == ''

"""world_size-2 gloo test of the N>1 path: contiguous sample sharding and the
single flat result gather (mmmot_amd/dist.py), without a GPU."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mmmot_amd.dist import flatten_results, gather_flat, gather_results, shard_range, unflatten_results


def fake_result(seed, N, M):
    g = torch.Generator().manual_seed(seed)
    L = N + M
    return (torch.rand(3, L, generator=g), [torch.rand(3, N, M, generator=g)], torch.rand(3, L, generator=g),
            torch.rand(3, L, generator=g))


def test_shard_range_is_a_partition():
    for B in (0, 1, 7, 8, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_range(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_flatten_roundtrip():
    res = [fake_result(1, 3, 4), fake_result(2, 1, 1)]
    flat, layout = flatten_results(res)
    back = unflatten_results(flat, layout)
    for a, b in zip(res, back):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1][0], b[1][0]) and torch.equal(a[3], b[3])


def _worker(rank, world, port, B, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    shapes = [(2 + i % 3, 1 + i % 4) for i in range(B)]
    lo, hi = shard_range(B, rank, world)
    local = [fake_result(100 + i, *shapes[i]) for i in range(lo, hi)]
    every = gather_results(local)                      # ragged shapes: object exchange + flat gather
    ok = len(every) == B
    for i, r in enumerate(every):
        exp = fake_result(100 + i, *shapes[i])
        ok = ok and torch.equal(r[0], exp[0]) and torch.equal(r[1][0], exp[1][0]) and torch.equal(r[2], exp[2])
    same = [fake_result(7 + rank * 10 + i, 4, 5) for i in range(2)]
    every2 = gather_results(same, same_layout=True)    # benchmark path: no object exchange
    ok = ok and len(every2) == 2 * world and torch.equal(every2[2 * rank][0], same[0][0])
    flats = gather_flat(torch.arange(rank + 1, dtype=torch.float32))
    ok = ok and [f.numel() for f in flats] == list(range(1, world + 1))
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_two_rank_gather_gloo():
    world, B = 2, 5
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, B, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_uneven_shards_cover_the_batch_exactly():
    """B = 250 over 8 ranks (cfg4's 256 minus a few): contiguous, balanced, no pair lost or doubled"""
    for B, world in ((250, 8), (7, 8), (256, 8), (1, 2), (33, 4)):
        bounds = [shard_range(B, r, world) for r in range(world)]
        assert bounds[0][0] == 0 and bounds[-1][1] == B
        assert all(bounds[r][1] == bounds[r + 1][0] for r in range(world - 1))
        sizes = [hi - lo for lo, hi in bounds]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_bench_single_command_launcher_dry_run():
    """`python bench.py --gpus 2` must become two ranks by itself (the driver launches it exactly so): the --dry
    mode runs bench.py's own launcher, shard_range split, gather_results(same_layout=True) loop, barrier /
    max-over-ranks timing and rank-0-only JSON on CPU tensors over gloo with a stub step."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--dry', '--gpus', '2', '--workload', 'tiny',
                        '--steps', '3', '--warmup', '1', '--pairs', '3'], capture_output=True, text=True, timeout=300,
                       env=env, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = p.stdout.splitlines()
    # rank 0 only, and NOTHING else on stdout: gloo's rank banners (like RCCL's version banner on the device) go to stderr -
    # bench.claim_stdout() - because the driver parses the last stdout line
    assert len(lines) == 1 and lines[0].startswith('{'), p.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 3 and out['warmup'] == 1 and out['gather_ok'] is True
    assert out['scaling'] == 'weak' and out['config']['pairs_per_step_per_gpu'] == 3
    # host placement / telemetry of every rank: one entry per rank; no GPU here, so no NUMA node is
    # found, nothing is bound and the clock / power samples are empty - the fields and the rank gather are what is checked
    pr = out['per_rank']
    assert set(pr) >= {'ms_per_step', 'pairs', 'sclk_mhz', 'sclk_vs_solo', 'power_w', 'numa_node', 'cpus', 'bound'}
    assert all(len(v) == 2 for v in pr.values()) and pr['bound'] == [False, False] and all(c >= 1 for c in pr['cpus'])
    assert len(lines[0]) <= 6000 and 'solo' in out and 'roofline' in out and 'cpu_baseline' in out


def test_bench_eight_ranks_uneven_shards_dry_run():
    """`python bench.py --gpus 8 --global-pairs 250` (cfg4's 256 minus a few): eight gloo ranks, shards of 32 / 31 pairs,
    the RAGGED form of the result gather; every rank checks that the gathered list is the whole batch in global pair
    order; rank 0's line carries one per_rank entry per rank (pairs, first_pair, the clock-ratio field) and stays
    under the size limit with every table populated."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['OMP_NUM_THREADS'] = '1'
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--dry', '--gpus', '8', '--workload', 'tiny',
                        '--steps', '2', '--warmup', '1', '--global-pairs', '250'], capture_output=True, text=True,
                       timeout=600, env=env, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = p.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith('{') and len(lines[0]) <= 6000, p.stdout[:2000]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 8 and out['config']['global_pairs'] == 250
    pr = out['per_rank']
    assert pr['pairs'] == [32, 32, 31, 31, 31, 31, 31, 31] and sum(pr['pairs']) == 250
    assert pr['first_pair'] == [0, 32, 64, 95, 126, 157, 188, 219]
    assert all(len(v) == 8 for v in pr.values()) and len(pr['sclk_vs_solo']) == 8
    det = [l for l in p.stderr.splitlines() if l.startswith('BENCH_DETAIL ')]
    assert len(det) == 1 and json.loads(det[0][len('BENCH_DETAIL '):])['gather_ok'] is True


def test_numa_binding_follows_the_gpu_sysfs_entries(tmp_path, monkeypatch):
    """bench.py binds a rank to the cores sysfs lists as local to its GPU (numa_node / local_cpulist of the PCI
    device); node -1 or a missing entry leaves the affinity alone"""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench._cpulist('0-3,8,10-11') == {0, 1, 2, 3, 8, 10, 11} and bench._cpulist('') == set()
    have = sorted(os.sched_getaffinity(0))
    d = tmp_path / 'dev'
    d.mkdir()
    (d / 'numa_node').write_text('1\n')
    (d / 'local_cpulist').write_text('%d\n' % have[-1])
    monkeypatch.setattr(bench, 'gpu_sysfs_dir', lambda i: str(d))
    try:
        info = bench.bind_to_gpu_numa_node(0, enable=False)
        assert info == {'numa_node': 1, 'cpus': len(have), 'bound': False}
        info = bench.bind_to_gpu_numa_node(0)
        assert info == {'numa_node': 1, 'cpus': 1, 'bound': True} and os.sched_getaffinity(0) == {have[-1]}
        os.sched_setaffinity(0, have)
        (d / 'numa_node').write_text('-1\n')
        assert bench.bind_to_gpu_numa_node(0)['bound'] is False and os.sched_getaffinity(0) == set(have)
    finally:
        os.sched_setaffinity(0, have)


def test_forced_one_rank_gather_runs_the_collectives():
    """``force=True`` (bench.py --force-dist, tests/test_rccl_gpu.py): a one-rank group still packs, all-gathers and
    unpacks - the call sequence of the N-rank step"""
    import socket
    sk = socket.socket()
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
    sk.close()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1')
    dist.init_process_group('gloo', rank=0, world_size=1)
    try:
        calls = []
        orig = dist.all_gather_into_tensor
        dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        try:
            res = [fake_result(5, 3, 4), fake_result(6, 3, 4)]
            assert gather_results(res, same_layout=True) is not res and not calls   # default: no collective
            got = gather_results(res, same_layout=True, force=True)
            assert len(calls) == 1 and all(torch.equal(a[1][0], b[1][0]) for a, b in zip(got, res))
            rag = gather_flat(torch.arange(7, dtype=torch.float32), equal=False, force=True)
            assert len(calls) == 3 and torch.equal(rag[0], torch.arange(7, dtype=torch.float32))
        finally:
            dist.all_gather_into_tensor = orig
    finally:
        dist.destroy_process_group()


def test_executed_flops_figures_quoted_in_the_design_notes():
    """F_ref / F_exec per frame pair (bench.py; DESIGN.md sections 5 and 7 quote them): reference-as-written FLOPs of
    SURVEY 8d, and the FLOPs of the math this build executes - each GEMM once, plus the two Gram matrices"""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_mod2', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    P = 128 * 2048
    assert round(bench.reference_flops_per_pair(64, 64, 128, P, 'C') / 1e9, 1) == 1824.6
    assert round(bench.executed_flops_per_pair(64, 64, 128, P, 'C') / 1e9, 1) == 1412.3
    assert round(bench.executed_flops_per_pair(64, 64, 128, P, 'C', rows=(1,)) / 1e9, 1) == 114.8
    assert round(bench.executed_flops_per_pair(64, 64, 128, P, 'C', rows=(0,)) / 1e9, 1) == 1290.1
    # per point: one pass of every layer (184 521 MAC) + the 64 x 64 and 128 x 128 Gram matrices
    per_pt = (bench.executed_flops_per_pair(64, 64, 128, P + 1000, 'C') - bench.executed_flops_per_pair(64, 64, 128, P, 'C')) / 2000
    assert per_pt == 184521 + 4096 + 16384

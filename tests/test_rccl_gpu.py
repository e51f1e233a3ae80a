"""The RCCL branch of the N-GPU step on ONE GPU (it had never executed anywhere): a one-rank
``nccl`` process group on the device, then the collective sequence of ``bench.py --gpus N`` - barrier, the flat
``all_gather_into_tensor`` of the scores in both forms (equal / ragged lengths), the python-object layout exchange,
the max-over-ranks all_reduce of the step time - on device tensors.  No scaling claim."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from mmmot_amd.dist import gather_flat, gather_results

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def one_rank_group():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', device_id=dev)
    yield dev
    dist.destroy_process_group()


def test_flat_gather_and_result_gather_over_rccl(one_rank_group):
    dev = one_rank_group
    assert dist.get_backend() == 'nccl' and dist.get_world_size() == 1
    g = torch.Generator().manual_seed(0)
    res = [(torch.rand(3, 13, generator=g).to(dev), [torch.rand(3, 6, 7, generator=g).to(dev)],
            torch.rand(3, 13, generator=g).to(dev), torch.rand(3, 13, generator=g).to(dev)) for _ in range(4)]
    for same in (True, False):   # benchmark form (one data collective) and ragged form (+ length / layout exchange)
        got = gather_results(res, same_layout=same, force=True)
        assert len(got) == 4
        for a, b in zip(got, res):
            assert a[0].is_cuda and torch.equal(a[0], b[0]) and torch.equal(a[1][0], b[1][0])
            assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    flat = torch.arange(1000, dtype=torch.float32, device=dev)
    for equal in (True, False):
        out = gather_flat(flat, equal=equal, force=True)
        assert len(out) == 1 and torch.equal(out[0], flat)
    dist.barrier()
    t = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == 1.25


def test_forward_results_through_the_collective(one_rank_group):
    """scores of a real forward_batch travel through the forced gather unchanged"""
    from common import build_model, get_case
    from mmmot_amd.synth import make_pair
    dev = one_rank_group
    c, base = get_case('s2_C_minus_abs_dual_add')
    m = build_model(c, base, device=dev)
    ins = [make_pair(c['N'], c['M'], c['S'], c['pts'], seed=900 + i, ragged=True) for i in range(3)]
    plan = m.make_plan([([c['N'], c['M']], x[1]['points_split'].reshape(-1).long().numpy()) for x in ins], c['S'])
    crops = torch.cat([x[0] for x in ins]).to(dev)
    points = torch.cat([x[1]['points'].reshape(-1, 3) for x in ins]).to(dev)
    with torch.no_grad():
        res = m.forward_batch(plan, crops, points)
    got = gather_results(res, same_layout=True, force=True)
    for a, b in zip(got, res):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1][0], b[1][0]) and torch.equal(a[3], b[3])


def test_uneven_shards_of_250_pairs_over_8_ranks_through_the_ragged_gather(one_rank_group):
    """B = 250 frame pairs over 8 ranks (shard_range: 32, 32, 31, ...): every rank's shard - ragged N x M per pair -
    travels through gather_results(same_layout=False) on the forced one-rank RCCL group and comes back unchanged; the
    shards tile [0, 250) exactly"""
    from mmmot_amd.dist import shard_range
    dev = one_rank_group
    B, world = 250, 8
    bounds = [shard_range(B, r, world) for r in range(world)]
    assert bounds[0][0] == 0 and bounds[-1][1] == B and all(bounds[r][1] == bounds[r + 1][0] for r in range(world - 1))
    assert sorted(hi - lo for lo, hi in bounds) == [31] * 6 + [32] * 2
    g = torch.Generator().manual_seed(5)
    for r in (0, 2, 7):   # a 32-pair shard, a 31-pair shard, the last one
        lo, hi = bounds[r]
        res = []
        for i in range(lo, hi):
            n, m = 1 + i % 9, 1 + (3 * i) % 7
            res.append((torch.rand(3, n + m, generator=g).to(dev), [torch.rand(3, n, m, generator=g).to(dev)],
                        torch.rand(3, n + m, generator=g).to(dev), torch.rand(3, n + m, generator=g).to(dev)))
        got = gather_results(res, same_layout=False, force=True)
        assert len(got) == hi - lo
        for a, b in zip(got, res):
            assert torch.equal(a[0], b[0]) and torch.equal(a[1][0], b[1][0]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])


def test_bench_forced_dist_cfg4_end_to_end():
    """`python bench.py --gpus 1 --force-dist --workload cfg4`: the N-GPU step (nccl process group, barrier, forward of
    the rank's shard, flat all_gather of the scores, max-over-ranks timing) end to end on one GPU; the JSON carries every
    rank's step time and the device time of its gather"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--force-dist', '--workload', 'cfg4',
                        '--pairs', '4', '--steps', '2', '--warmup', '1', '--cpu-pairs', '0', '--extra-trunks', 'none',
                        '--no-latency', '--no-workloads'], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][-1])
    assert out['config']['rccl'] is True and out['n_gpus'] == 1 and out['value'] > 0
    assert out['parity']['linf_vs_reference_golden'] < 1e-3
    pr = out['per_rank']
    assert len(pr['ms_per_step']) == 1 and pr['ms_per_step'][0] > 0 and pr['gather_us_per_step'][0] > 0

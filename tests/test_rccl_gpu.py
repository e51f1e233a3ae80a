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

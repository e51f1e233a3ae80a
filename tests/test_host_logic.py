"""Host logic (weight packing / folding, batch plan tables, launch schedule)
validated WITHOUT a GPU: the engine runs over tests/fake_ops.TorchOps - a
test-only executable statement of the C-ABI contracts - and must reproduce the
reference golden vectors.  On the GPU box the same engine code runs over
HipOps (tests/test_parity_gpu.py)."""
import numpy as np
import pytest
import torch

from common import TOL, build_model, case_inputs, case_names, compare_outputs, get_case, golden
from fake_ops import TorchOps
from mmmot_amd import TrackingNet, build_model as build_from_config, model_kwargs_from_config
from mmmot_amd.plan import BatchPlan, RowTiles
from mmmot_amd.weights import gen_tensor


@pytest.mark.parametrize('trunk,tol', [('f16q8', 5e-4), ('f16x3', 2e-4)])
@pytest.mark.parametrize('name', case_names(light_only=True))
def test_engine_schedule_matches_golden(name, trunk, tol):
    """both trunk arithmetics (default f16q8: correction terms in e4m3; f16x3: fp32-class) through the emulated C-ABI"""
    c, base = get_case(name)
    m = build_model(c, base, ops=TorchOps())
    m.set_trunk(trunk)
    m.engine().q8_min_crop = 0  # f16q8: the e4m3 arithmetic on every crop size of the fixtures
    out = m(*case_inputs(c))
    compare_outputs(out, golden(name), tol=tol)


def test_batched_plan_equals_single_samples():
    """B samples in one plan (ragged sizes) == each sample alone: statistics never cross samples."""
    ca, base = get_case('s2_C_minus_abs_dual_add')
    cb, _ = get_case('s1_C_minus_abs_dual_add')
    m = build_model(ca, base, ops=TorchOps())
    ins = [case_inputs(ca), case_inputs(dict(cb, S=ca['S']))]
    singles = [m(*x) for x in ins]
    samples = [([int(d) for d in ds], info['points_split'].reshape(-1).long().numpy()) for _, info, ds in ins]
    plan = m.make_plan(samples, ca['S'])
    crops = torch.cat([x[0] for x in ins])
    points = torch.cat([x[1]['points'].reshape(-1, 3) for x in ins])
    batch = m.forward_batch(plan, crops, points)
    for (det, links, new, end), (sdet, slinks, snew, send, _) in zip(batch, singles):
        assert torch.allclose(det, sdet, atol=1e-6) and torch.allclose(new, snew, atol=1e-6)
        assert torch.allclose(end, send, atol=1e-6) and torch.allclose(links[0], slinks[0], atol=1e-6)


def test_single_modality_rows():
    c, base = get_case('s2_C_multiply_none')
    m = build_model(c, base, ops=TorchOps())
    m.set_trunk('f16x3')  # fp32-class arithmetic: the tolerance below checks the row logic, not the trunk
    dets, info, ds = case_inputs(c)
    g = golden(c['name'])
    out0 = m.forward_rows(dets, info, ds, rows=(0,))
    compare_outputs(out0, g, tol=2e-4, rows=(0,))
    out1 = m.forward_rows(None, info, ds, rows=(1,))
    compare_outputs(out1, g, tol=2e-4, rows=(1,))


@pytest.mark.parametrize('fusion', ['A', 'B', 'C'])
def test_state_dict_keys_and_shapes_equal_the_reference(fusion):
    """Exact key -> shape equality with the IMPORTED reference's ``TrackingNet.state_dict()`` (manifest written by
    oracle/gen_golden.py from /root/reference): same keys, same order, same shapes."""
    import json
    import os
    from common import GOLD
    with open(os.path.join(GOLD, 'state_dict_manifest.json')) as f:
        man = json.load(f)
    ref = man['keys'][fusion]
    m = TrackingNet(**dict(man['base_kwargs'], score_fusion_arch=fusion, affinity_op='multiply', softmax_mode='none'))
    ours = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert list(ours) == list(ref), (sorted(set(ours) ^ set(ref))[:8])
    assert ours == ref
    assert sum(int(np.prod(v)) for v in ref.values()) == {'A': 20691364, 'B': 20692900, 'C': 21218212}[fusion]


def test_build_model_from_reference_yaml_dict():
    common = dict(sample_max_len=2, without_reflectivity=True, dropblock=0, use_dropout=False,
                  model=dict(point_arch='v1', point_len=512, appear_arch='vgg', appear_len=512, appear_skippool=True,
                             appear_fpn=False, end_arch='v2', end_mode='avg', affinity_op='minus_abs',
                             softmax_mode='dual_add', score_arch='branch_cls', neg_threshold=0.2,
                             score_fusion_arch='C', test_mode=2))
    m = build_from_config({'common': common})
    assert isinstance(m, TrackingNet) and m.test_mode == 2 and m.affinity_op == 'minus_abs'
    assert model_kwargs_from_config(common)['seq_len'] == 2


def test_modes_and_no_cpu_fallback():
    c, base = get_case('s1_A_multiply_none')
    m = build_model(c, base)
    # train() / eval() record the flag (the reference toggles them around validation, tracking_model.py:32-48); the
    # batched / single-modality entry points are eval-only, a training-mode forward goes to mmmot_amd/train.py - and, like
    # every path, refuses CPU tensors with the default backend
    assert m.train() is m and m.training
    with pytest.raises(NotImplementedError):
        m.forward_rows(*case_inputs(c), rows=(0,))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m(*case_inputs(c))
    assert m.eval() is m and not m.training
    # default backend = HipOps: CPU tensors must be refused, never computed on the host
    with pytest.raises(RuntimeError):
        m(*case_inputs(c))


def test_plan_tables():
    t = RowTiles([5, 300, 128], 'cpu')
    assert t.T == 1 + 3 + 1 and t.R == 433
    assert list(t.h_nrows) == [5, 128, 128, 44, 128] and list(t.h_group) == [0, 1, 1, 1, 2]
    assert list(t.h_g_tile0) == [0, 1, 4] and list(t.h_row0) == [0, 5, 133, 261, 305]
    ps = np.array([0, 3, 4, 10, 12, 20])
    p = BatchPlan([([2, 3], ps)], 32, 'cpu')
    assert p.Lt == 5 and p.P == 20 and p.pairs == [(0, 0, 2, 2, 3)]
    assert list(p.row_det.numpy()) == [0] * 3 + [1] + [2] * 6 + [3] * 2 + [4] * 8
    # new rows scatter to current-frame detections, end rows to previous-frame ones, per modality row
    om = p.v_omap.numpy().reshape(3, 5)
    assert list(om[0]) == [2, 3, 4, 15 + 0, 15 + 1] and list(om[1]) == [7, 8, 9, 20, 21]
    with pytest.raises(ValueError):
        BatchPlan([([2, 0], ps)], 32, 'cpu')
    with pytest.raises(ValueError):
        BatchPlan([([2, 3], np.array([0, 3, 3, 10, 12, 20]))], 32, 'cpu')  # empty detection


def test_weight_generator_is_order_independent():
    a = gen_tensor('w_det.0.weight', (512, 512, 1), 0)
    b = gen_tensor('w_det.0.weight', (512, 512, 1), 0)
    assert np.array_equal(a, b) and not np.array_equal(a, gen_tensor('w_det.3.weight', (512, 512, 1), 0))
    assert np.array_equal(gen_tensor('x.idt', (3, 3)), np.eye(3, dtype=np.float32))


def test_custom_operators_are_registered_for_the_device_only():
    """mmmot_amd/torch_ops.py: the module API reaches the HIP library through torch.library operators with a CUDA
    (= HIP) kernel and a Meta kernel - and no CPU kernel (no fallback)."""
    import torch
    from mmmot_amd import torch_ops
    c, base = get_case('s2_C_multiply_none')
    m = build_model(c, base)
    eng = m.engine()
    assert eng.ops.name == 'hip'
    dets, info, ds = case_inputs(c)
    L, P = dets.shape[0], info['points'].shape[1]
    plan = m.make_plan([([c['N'], c['M']], info['points_split'].reshape(-1).long().numpy())], c['S'])
    eh, ph = torch_ops.engine_handle(eng), torch_ops.plan_handle(plan)
    for name in ('forward_batch', 'appearance', 'pointnet'):
        assert hasattr(torch.ops.mmmot, name)
    # Meta kernels: shapes without touching a device (FakeTensor / tracing)
    det, link, new, end = torch.ops.mmmot.forward_batch(torch.empty(L, 3, c['S'], c['S'], device='meta'),
                                                        torch.empty(P, 3, device='meta'), eh, ph)
    assert det.shape == (3, L) and new.shape == (3, L) and end.shape == (3, L) and link.shape == (3 * c['N'] * c['M'],)
    assert det.device.type == 'meta'
    assert torch.ops.mmmot.appearance(torch.empty(L, 3, c['S'], c['S'], device='meta'), eh, ph).shape == (L, 512)
    # CPU tensors: the dispatcher has no kernel to run
    with pytest.raises(NotImplementedError):
        torch.ops.mmmot.forward_batch(dets, info['points'].reshape(-1, 3), eh, ph)
    # stale handles are refused
    with pytest.raises(RuntimeError):
        torch.ops.mmmot.forward_batch(torch.empty(1, device='meta'), None, 10 ** 9, ph)


def test_refresh_head_copies_in_place_and_parameter_edits_are_noticed():
    """a captured graph holds raw pointers into the packed head; refresh_head() must keep the addresses
    (in-place copy) and the engine must notice in-place parameter edits (optimizer.step) through the version counters"""
    c, base = get_case('s2_C_multiply_none')
    m = build_model(c, base, ops=TorchOps())
    eng = m.engine()
    ptrs = {k: v.data_ptr() for k, v in eng.P['w_link'].items() if torch.is_tensor(v)}
    assert m.head_is_current()
    before = m(*case_inputs(c))[1][0].clone()
    with torch.no_grad():
        m.w_link.conv1[3].weight.mul_(1.001)
    assert not m.head_is_current()
    assert m.refresh_head() is eng and m.head_is_current()
    assert {k: v.data_ptr() for k, v in eng.P['w_link'].items() if torch.is_tensor(v)} == ptrs
    after = m(*case_inputs(c))[1][0]
    assert (after - before).abs().max() > 0      # the new weights are the ones that ran
    v = m._pack_version
    m.invalidate()
    assert m._pack_version == v + 1              # replaced packs mark captured graphs stale


def test_engine_refuses_concurrent_forwards():
    """the workspace arena is shared mutable state behind an integer handle"""
    c, base = get_case('s2_C_multiply_none')
    m = build_model(c, base, ops=TorchOps())
    eng = m.engine()
    assert eng._busy.acquire(blocking=False)
    try:
        with pytest.raises(RuntimeError, match='concurrent forwards'):
            m(*case_inputs(c))
    finally:
        eng._busy.release()
    m(*case_inputs(c))


def test_crop_sides_follow_the_reference_pooling():
    """any even crop side >= 32 runs (odd maps on the way down are floored like nn.MaxPool2d(2, 2), goldens s8_*);
    an odd side or one that leaves nothing after five poolings is refused with a clear error"""
    c, base = get_case('s8_S40_C')
    m = build_model(c, base, ops=TorchOps())
    dets, info, ds = case_inputs(c)
    for bad in (30, 41):
        with pytest.raises(ValueError, match='crop side'):
            m(torch.zeros(dets.shape[0], 3, bad, bad), info, ds)


@pytest.mark.parametrize('trunk', ['f16x3', 'f32'])
def test_uint8_crops_are_the_same_forward_as_the_normalised_tensor(trunk):
    """the forward takes the 8-bit crops of the resize; ToTensor / Normalize then run on the device with
    the reference's fp32 arithmetic (inside the fused first trunk launch, or as one small kernel for the fp32 trunk)"""
    from mmmot_amd.crops import MEAN, STD
    c, base = get_case('s2_C_multiply_none')
    m = build_model(c, base, ops=TorchOps())
    m.set_trunk(trunk)
    dets, info, ds = case_inputs(c)
    g = torch.Generator().manual_seed(1)
    u8 = torch.randint(0, 256, (dets.shape[0], c['S'], c['S'], 3), generator=g, dtype=torch.uint8)
    ref_in = TorchOps._normalize_u8(u8, MEAN, STD)
    a = m(ref_in, info, ds)
    b = m(u8, info, ds)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1][0], b[1][0]) and torch.equal(a[3], b[3])
    with pytest.raises(ValueError, match='crops must be'):
        m(u8.permute(0, 3, 1, 2).contiguous(), info, ds)


def test_image_first_token_semantics():
    """Engine.image_first issues the image branch of the NEXT forward: that forward (same crops tensor, same geometry)
    skips its own image branch and returns the same scores; the token is consumed by one forward and never taken for
    other crops."""
    c, base = get_case('s2_C_minus_abs_dual_add')
    m = build_model(c, base, ops=TorchOps())
    dets, info, ds = case_inputs(c)
    fc = [int(d) for d in ds]
    ps = info['points_split'].reshape(-1).long().numpy()
    S = int(dets.shape[-1])
    plan = m.make_plan([(fc, ps)], S)
    plan_img = m.make_plan([(fc, None)], S, rows=(0,))
    crops = dets.contiguous()
    points = info['points'].reshape(-1, 3).contiguous()
    eng = m.engine()
    with torch.no_grad():
        want = {k: v.clone() for k, v in eng.forward(plan, crops, points).items() if k in ('det', 'link', 'new', 'end')}
        calls = []
        orig = eng._guarded_appearance
        eng._guarded_appearance = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        eng.image_first(plan_img, crops)
        assert len(calls) == 1 and eng._image_token is not None
        got = eng.forward(plan, crops, points)
        assert len(calls) == 1 and eng._image_token is None          # skipped its own image branch, token consumed
        for k in want:
            assert torch.equal(got[k], want[k]), k
        eng.forward(plan, crops, points)
        assert len(calls) == 2                                        # the next forward runs it again
        eng.image_first(plan_img, crops)
        other = crops.clone()
        eng.forward(plan, other, points)
        assert len(calls) == 4 and eng._image_token is None          # other tensor: not taken, and dropped


def test_conv1_weight_shift_keeps_weights_and_bias_inside_fp16():
    """The fused first layer carries its folded bias in the k = 27 slot of the fp16 hi/lo weight records
    (csrc/conv3x3_hl16_patch.hip): one power-of-two scale has to keep BOTH inside fp16's range."""
    from mmmot_amd.pack import conv1_weight_shift, hl16_weight_shift
    g = torch.Generator().manual_seed(5)
    for wscale, bscale in ((0.3, 0.1), (0.3, 40.0), (1e-3, 5.0), (20.0, 0.0), (0.5, 1e-6)):
        w = torch.randn(64, 32, generator=g, dtype=torch.float64) * wscale
        b = torch.randn(64, generator=g, dtype=torch.float64) * bscale
        s = conv1_weight_shift(w, b)
        assert s <= hl16_weight_shift(w)
        assert float(w.abs().max()) * 2.0 ** s <= 32768.0
        assert float(b.abs().max()) * 2.0 ** s <= 32768.0 < 65504.0
        if bscale == 0.0:
            assert s == hl16_weight_shift(w)


def test_point_gather_staging_buffer_is_per_thread(monkeypatch):
    """the pinned table-upload buffer of gather_points_batched is filled on the host and copied
    asynchronously; only the calling thread's own stream synchronisation makes its reuse safe, so every thread must own
    its buffer (pin_memory needs a device: replaced by a plain allocation here)."""
    import threading
    from mmmot_amd import points as P
    monkeypatch.setattr(torch.Tensor, 'pin_memory', lambda self: self, raising=False)
    got = {}

    def grab(name):
        a = P._staging(100)
        got[name] = (a, P._staging(50), P._staging(10 * a.numel()))

    ts = [threading.Thread(target=grab, args=(n,)) for n in ('t0', 't1')]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    grab('main')
    bufs = [got[n][0].data_ptr() for n in ('t0', 't1', 'main')]
    assert len(set(bufs)) == 3                      # one buffer per thread
    for a, again, grown in got.values():
        assert again.data_ptr() == a.data_ptr()     # reused while it is large enough
        assert grown.numel() >= 10 * a.numel()      # grown (replaced) only for its own thread


def test_launch_profiler_classifies_every_forward_launch():
    """bench.py's extra.kernels: every operator call of a forward lands in a launch class with its algorithmic FLOPs and
    bytes (cost models of mmmot_amd/profiler.py run over the torch emulation of the C-ABI; timing needs the device)"""
    from mmmot_amd.profiler import COSTS, LaunchProfiler
    c, base = get_case('s2_C_minus_abs_dual_add')
    m = build_model(c, base, ops=TorchOps())
    ops = m.engine().ops
    before = dict(ops.__dict__)
    with torch.no_grad(), LaunchProfiler(ops) as prof:
        m(*case_inputs(c))
    assert dict(ops.__dict__).keys() == before.keys()  # the wrappers are gone
    s = prof.summary(steps=1)
    labels = {r['class'] for r in s['classes']}
    assert not labels & set(COSTS), labels  # a bare method name = a cost model that failed to parse its arguments
    by = {r['class']: r for r in s['classes']}
    L, S = 12, 64
    conv2 = by['trunk conv 64->128 @32x32']
    assert conv2['gflop_per_step'] == round(2.0 * L * 32 * 32 * 9 * 64 * 128 / 1e9, 1)
    fused = [r for r in s['classes'] if r['class'].startswith('trunk conv1_1+conv1_2')][0]
    assert fused['gflop_per_step'] == round(2.0 * L * S * S * 9 * (3 * 64 + 64 * 64) / 1e9, 1)
    assert any(r['class'].startswith('A-resident GEMM 128->1024') for r in s['classes'])
    assert any('pair prologue' in r['class'] for r in s['classes'])
    assert sum(r['launches_per_step'] for r in s['classes']) == len(prof.records)


def test_gram_route_with_per_detection_bias_specification():
    """the float64 specification of mmmot_gn_finalize_gram_dbias (tests/fake_ops.py) against direct statistics of
    v = W relu(gn(x)) + dbias[det] - ragged detections, several super-tiles per detection, two samples"""
    emu = TorchOps(torch.float64)
    K, N = 64, 96
    dets = [[300, 1, 130], [90, 257]]
    counts = [sum(d) for d in dets]
    tiles = RowTiles(counts, 'cpu', sub_counts=dets, tile=128)
    Lt = sum(len(d) for d in dets)
    tile_det = torch.from_numpy(np.repeat(np.arange(Lt), tiles.h_sub_ntiles)).int()
    g = torch.Generator().manual_seed(11)
    X = torch.randn(sum(counts), K, generator=g) * 2 + 0.7
    sc, sh = torch.rand(2, K, generator=g) + 0.5, torch.randn(2, K, generator=g) * 0.5
    W = torch.randn(N, K, generator=g) * K ** -0.5
    dbias = torch.randn(Lt, N, generator=g) * 3 + 1.5
    gamma, beta = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g)
    grp = torch.repeat_interleave(torch.arange(2), torch.tensor(counts))
    row_det = torch.repeat_interleave(torch.arange(Lt), torch.tensor([c for d in dets for c in d]))
    v = torch.relu(X.double() * sc.double()[grp] + sh.double()[grp]) @ W.double().t() + dbias.double()[row_det]
    Gp, Sp = torch.zeros(tiles.T, K * K, dtype=torch.float64), torch.zeros(tiles.T, K, dtype=torch.float64)
    emu.gram_rows(X, K, sc, sh, tiles, Gp, Sp)
    s_, h_ = torch.zeros(2, N), torch.zeros(2, N)
    emu.gn_finalize_gram_dbias(Gp, Sp, tiles, tile_det, K, W, dbias, N, gamma, beta, 1e-5, None, s_, h_)
    for gi in range(2):
        vg = v[grp == gi]
        want = gamma.double() / torch.sqrt(vg.var(0, unbiased=False) + 1e-5)
        assert (s_[gi].double() - want).abs().max().item() < 1e-6
        assert (h_[gi].double() - (beta.double() - vg.mean(0) * want)).abs().max().item() < 1e-5

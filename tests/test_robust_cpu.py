"""Host logic of the trunk range guard (mmmot_amd/engine.py) and the trained-like weight profiles, on the torch
emulation of the C-ABI (no GPU): the engine must notice activations beyond the e4m3 / fp16 range of the hq8 / hl16
formats and lower the trunk arithmetic f16q8 -> f16x3 -> f32, so that the outputs stay inside the 1e-3 budget."""
import warnings

import numpy as np
import pytest
import torch

from fake_ops import TorchOps
from mmmot_amd import TrackingNet
from mmmot_amd.pack import hl16_channel_shifts, pack_weights
from mmmot_amd.synth import make_pair
from mmmot_amd.weights import calibrate_bn, generate_state_dict_trained
from oracle import restatement as R

KW = dict(seq_len=2, score_arch='branch_cls', appear_arch='vgg', appear_len=512, appear_skippool=True, appear_fpn=False,
          point_arch='v1', point_len=512, without_reflectivity=True, end_arch='v2', end_mode='avg', test_mode=2,
          neg_threshold=0.2, dropblock=0, use_dropout=False, score_fusion_arch='A', affinity_op='multiply',
          softmax_mode='none')
CFG = dict(fusion='A', affinity_op='multiply', softmax_mode='none', neg_threshold=0.2, score_arch='branch_cls')


def build(profile, seed, scale, S=64, trunk='f16q8'):
    m = TrackingNet(**KW)
    sd = generate_state_dict_trained(m.state_dict(), seed, profile)
    dets, info, ds = make_pair(2, 2, S, 12, seed=4000 + seed, ragged=True)
    dets = dets * scale
    if profile == 'calibrated':
        calibrate_bn(sd, make_pair(4, 4, S, 4, seed=4100 + seed)[0] * scale)
    m.load_state_dict(sd)
    m.eval()
    m.set_ops(TorchOps())
    m.set_trunk(trunk)
    return m, sd, (dets, info, ds)


def linf(out, ref):
    return max((out[0] - ref[0]).abs().max().item(), (out[1][0] - ref[1][0]).abs().max().item(),
               (out[2] - ref[2]).abs().max().item(), (out[3] - ref[3]).abs().max().item())


def test_wild_statistics_trip_the_guard_and_the_result_stays_in_budget():
    """uncalibrated BatchNorm statistics (running_var / gamma log-uniform over 4 / 2.5 decades): activations leave the
    fp16 range, the forced reduced-range arithmetic is wrong by O(1), the guarded engine ends in f32 and is right"""
    m, sd, (dets, info, ds) = build('wild', 0, 1.0)
    with torch.no_grad():
        ref = R.tracking_forward(sd, CFG, dets, info['points'], info['points_split'], [2, 2])
        with pytest.warns(RuntimeWarning, match='range guard'):
            out = m(dets, info, ds)
    eng = m.engine()
    assert eng.trunk == 'f32' and eng.trunk_requested == 'f16q8'
    assert eng.range_events and eng.range_events[-1]['now'] == 'f32' and eng.range_events[-1]['fp16_clamped'] > 0
    assert linf(out, ref) < 1e-3
    # sticky: the next forward runs f32 straight away, without another event
    n = len(eng.range_events)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter('error')
        out2 = m(dets, info, ds)
    assert len(eng.range_events) == n and linf(out2, ref) < 1e-3
    # with the guard off the same model is far outside the budget: the guard is what saves it
    m2, _, _ = build('wild', 0, 1.0)
    m2.engine().range_guard = False
    with torch.no_grad():
        bad = m2(dets, info, ds)
    assert linf(bad, ref) > 1e-2


def test_e4m3_saturation_alone_lowers_to_f16x3():
    """calibrated statistics with the inputs scaled so that activations cross 1792 but stay far below 65000: the e4m3
    copies saturate, the fp16 halves do not -> one step down, to f16x3"""
    m, sd, (dets, info, ds) = build('calibrated', 1, 1.0)
    eng = m.engine()
    eng.q8_layers = None
    # push one layer's activations over the e4m3 limit by scaling its folded bias / gain: emulate with a large
    # gamma on conv3_1's BatchNorm instead of touching the inputs (keeps the other layers in range)
    key = 'appearance.layers.1.1.weight'
    sd2 = dict(sd)
    sd2[key] = sd[key] * 0 + 900.0
    sd2['appearance.layers.1.4.weight'] = sd['appearance.layers.1.4.weight'] * 0 + 1e-3  # next layer scales back down
    m.load_state_dict(sd2)
    m.set_ops(TorchOps())
    m.set_trunk('f16q8')
    eng = m.engine()
    with torch.no_grad():
        ref = R.tracking_forward(sd2, CFG, dets, info['points'], info['points_split'], [2, 2])
        with pytest.warns(RuntimeWarning, match='range guard'):
            out = m(dets, info, ds)
    assert eng.trunk == 'f16x3', eng.range_events
    ev = eng.range_events[0]
    assert ev['was'] == 'f16q8' and ev['e4m3_saturated'] > 0 and ev['fp16_clamped'] == 0
    assert linf(out, ref) < 1e-3


def test_calibrated_statistics_do_not_trip_the_guard():
    m, sd, (dets, info, ds) = build('calibrated', 0, 1.0)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter('error')
        ref = R.tracking_forward(sd, CFG, dets, info['points'], info['points_split'], [2, 2])
        out = m(dets, info, ds)
    assert m.engine().trunk == 'f16q8' and not m.engine().range_events
    assert linf(out, ref) < 1e-3


def test_guard_checks_synchronously_on_the_first_forward_and_on_request():
    m, sd, ins = build('calibrated', 0, 1.0)
    eng = m.engine()
    eng.range_check_every = 3
    reads = []
    orig = eng.read_range
    eng.read_range = lambda reset=True: (reads.append(eng._n_forward), orig(reset))[1]
    with torch.no_grad():
        for _ in range(7):
            m(*ins)
    # per synchronous check: one read that opens the window, one after the trunk
    assert reads == [0, 0, 3, 3, 6, 6]


def test_guard_detects_late_without_synchronising():
    """after the first forward the counters are read back asynchronously and inspected by the NEXT forward: a model that
    leaves the range later (here: the weights are swapped under a live engine) is lowered one step late, with an event
    that says the previous forward was affected; each engine counts in its own block"""
    m, sd, (dets, info, ds) = build('calibrated', 0, 1.0)
    eng = m.engine()
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter('error')
        m(dets, info, ds)
        m(dets, info, ds)
    assert not eng.range_events and eng.trunk == 'f16q8'
    other, _, _ = build('calibrated', 1, 1.0)  # a second engine on the same "device": its own counter block
    with torch.no_grad():
        other(dets, info, ds)
    # blow conv3_1's gain up in the PACKED weights of the live engine (no re-pack: _n_forward keeps counting)
    cv = eng.P['vgg'][4]
    cv['bias'] = cv['bias'] + 3000.0
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter('error')
        m(dets, info, ds)            # out of range, not yet noticed: no synchronous read on this forward
    assert not eng.range_events
    with torch.no_grad(), pytest.warns(RuntimeWarning, match='forwards 2..2 of this engine ran out of range'):
        m(dets, info, ds)            # the read-back of the previous forward is inspected first
    ev = eng.range_events[0]
    assert ev['recomputed'] is False and ev['was'] == 'f16q8' and ev['e4m3_saturated'] > 0
    assert ev['affected_forwards'] == (2, 2) and ev['affected_forward'] == 2 and ev['forward'] == 3
    assert eng.last_out_of_range_forward == 2 and eng.out_of_range_window == (2, 2)
    assert eng.trunk in ('f16x3', 'f32')
    assert not other.engine().range_events and other.engine().trunk == 'f16q8'


def test_late_read_back_reports_the_forwards_it_covers():
    """a read-back recorded behind forward k is inspected only once its copy has completed - maybe several
    forwards later, and no new one is queued meanwhile.  The event must name the forwards the counters cover (those since
    the previous read-back up to k), not "the forward before the inspection"; the forwards that ran while the copy was
    pending belong to the NEXT read-back."""
    m, sd, (dets, info, ds) = build('calibrated', 0, 1.0)
    eng = m.engine()

    class Slow:  # stands for a torch.cuda.Event whose copy has not completed yet
        done = False

        def query(self):
            return self.done

    with torch.no_grad():
        m(dets, info, ds)                      # forward 0: synchronous check
        m(dets, info, ds)                      # forward 1: read-back queued (covers 1..1)
        m(dets, info, ds)                      # forward 2: inspects it, queues its own (covers 2..2)
        assert eng._range_pending[1:] == (2, 2)
        slow = Slow()
        eng._range_pending = (slow,) + eng._range_pending[1:]
        cv = eng.P['vgg'][4]
        good = cv['bias']
        m(dets, info, ds)                      # forwards 3, 4: the copy is still in flight, nothing new is queued
        cv['bias'] = good + 3000.0             # forward 4 leaves the range
        m(dets, info, ds)
        cv['bias'] = good
        assert not eng.range_events and eng._range_pending[0] is slow
        slow.done = True
        m(dets, info, ds)                      # forward 5: 2..2 was fine; queues a read-back that covers 3..5
        assert not eng.range_events and eng._range_pending[1:] == (3, 5)
        with pytest.warns(RuntimeWarning, match='forwards 3..5'):
            m(dets, info, ds)                  # forward 6 inspects it
    ev = eng.range_events[0]
    assert ev['affected_forwards'] == (3, 5) and ev['forward'] == 6 and eng.out_of_range_window == (3, 5)


def test_synchronous_check_gives_the_superseded_read_back_its_verdict():
    """with range_check_every > 0 a synchronous check may come round while the previous asynchronous read-back
    has not completed.  The counters it reads (and resets) then cover forwards that were already returned: they get their
    verdict and an event naming them instead of being dropped without one."""
    m, sd, (dets, info, ds) = build('calibrated', 0, 1.0)
    eng = m.engine()
    eng.range_check_every = 4

    class Slow:
        def query(self):
            return False

    with torch.no_grad():
        m(dets, info, ds)                      # forward 0: synchronous
        m(dets, info, ds)                      # forward 1: read-back queued (1..1)
        eng._range_pending = (Slow(),) + eng._range_pending[1:]   # ... and never completes
        cv = eng.P['vgg'][4]
        good = cv['bias']
        cv['bias'] = good + 3000.0
        m(dets, info, ds)                      # forward 2 leaves the range; nothing new is queued
        cv['bias'] = good
        m(dets, info, ds)                      # forward 3
        assert not eng.range_events
        with pytest.warns(RuntimeWarning, match='forwards 1..3 of this engine ran out of range'):
            out = m(dets, info, ds)            # forward 4: the synchronous check reads what 1..3 left behind
    ev = eng.range_events[0]
    assert ev['recomputed'] is False and ev['affected_forwards'] == (1, 3) and ev['forward'] == 4
    assert eng.out_of_range_window == (1, 3) and eng.trunk in ('f16x3', 'f32')
    with torch.no_grad():
        ref = R.tracking_forward(sd, CFG, dets, info['points'], info['points_split'], [2, 2])
    assert linf(out, ref) < 1e-3               # forward 4 itself ran (in the lowered arithmetic) and is fine


def test_per_channel_shifts_follow_the_folded_gains():
    """every output channel of every trunk layer lands in (2^13, 2^14] after its own power-of-two scale, whatever its
    BatchNorm gain; the scale vector undoes it exactly"""
    m = TrackingNet(**KW)
    sd = generate_state_dict_trained(m.state_dict(), 0, 'wild')
    P = pack_weights(sd, 'A', 'cpu')
    spread = []
    for cv in P['vgg'][1:]:
        w = cv['wp'].double()
        sh = hl16_channel_shifts(w)
        assert torch.equal(sh, cv['wshift'])
        scaled = w.abs().amax(dim=(0, 2)) * torch.pow(2.0, sh.double())
        assert (scaled > 8192).all() and (scaled <= 16384).all()
        assert torch.equal(cv['oscale'].double(), torch.pow(2.0, -sh.double()))
        spread.append(int(sh.max() - sh.min()))
    assert min(spread) >= 10, spread  # >= 3 decades of per-channel gain in every layer of the 'wild' profile

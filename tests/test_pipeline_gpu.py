"""The pieces of SURVEY section 8 composed the way the reference's TestSequence / TrackingModule compose them
(dataset/test_seq_dataset.py:176-246 -> tracking_model.py:56-117): two synthetic frames (RGB image, LiDAR sweep,
calibration, detections) -> device point gather + device crop/resize/normalise -> TrackingNet.forward ->
packed hand-off to the solver.  Checked against the same chain built from the CPU oracles."""
import numpy as np
import pytest
import torch

from common import TOL
from mmmot_amd import TrackingNet
from mmmot_amd.crops import crop_resize_normalize
from mmmot_amd.points import prep_points
from mmmot_amd.tracker_glue import scores_for_solver
from mmmot_amd.weights import init_module
from oracle import crops_ref, points_ref
from oracle import restatement as R
from oracle.gen_golden_crops import frame as make_frame
from oracle.gen_golden_points import IMG_SHAPE, P2, R0, TR, make_scene

pytestmark = pytest.mark.gpu
KW = dict(seq_len=2, score_arch='branch_cls', appear_arch='vgg', appear_len=512, appear_skippool=True, appear_fpn=False,
          point_arch='v1', point_len=512, without_reflectivity=True, end_arch='v2', end_mode='avg', test_mode=2,
          neg_threshold=0.2, dropblock=0, use_dropout=False, score_fusion_arch='C', affinity_op='multiply',
          softmax_mode='none')


def test_two_frames_end_to_end():
    S = 64
    info = {'calib/R0_rect': R0, 'calib/Tr_velo_to_cam': TR, 'calib/P2': P2, 'img_shape': IMG_SHAPE}
    rect, Tr, P2f = (a.astype(np.float32) for a in (R0, TR, P2))
    frames = []
    for f, (seed, ndet) in enumerate(((31, 5), (32, 6))):
        img = make_frame(40 + f, int(IMG_SHAPE[0]), int(IMG_SHAPE[1]))
        sweep, dets = make_scene(seed, 6000, ndet, empty_boxes=1)
        dets['bbox'] = np.clip(dets['bbox'], [-20, -20, 10, 10], [1200, 340, 1260, 390])  # plausible, may leave the frame
        frames.append((img, sweep, dets))

    # ---- device chain -----------------------------------------------------------------------------------------
    crops_d, pts_d, split_d, dsplit = [], [], [0], []
    for img, sweep, dets in frames:
        crops_d.append(crop_resize_normalize(torch.from_numpy(img).cuda(), dets['bbox'], S))
        pc = prep_points(torch.from_numpy(sweep).cuda(), info, dets, without_reflectivity=True)
        pts_d.append(pc['points'])
        split_d += [split_d[-1] + s for s in pc['points_split'][1:]]
        dsplit.append(torch.tensor([len(dets['rotation_y'])]))
    det_info = {'points': torch.cat(pts_d).unsqueeze(0),
                'points_split': torch.tensor(split_d, dtype=torch.float32).unsqueeze(0).cuda()}
    model = TrackingNet(**KW)
    init_module(model, seed=0)
    model.eval().cuda()
    with torch.no_grad():
        det, links, new, end, _ = model(torch.cat(crops_d), det_info, dsplit)
    det_s, link_s, new_s, end_s = scores_for_solver(det, links, new, end, model.test_mode)
    assert not det_s.is_cuda and link_s[0].shape == (1, 5, 6)

    # ---- oracle chain -------------------------------------------------------------------------------------------
    crops_o, pts_o, split_o = [], [], [0]
    for img, sweep, dets in frames:
        crops_o.append(crops_ref.crop_resize_normalize(img, dets['bbox'], S)[1])
        pc = points_ref.prep_points(sweep, rect, Tr, P2f, IMG_SHAPE, dets, without_reflectivity=True)
        pts_o.append(np.asarray(pc['points'], dtype=np.float32))
        split_o += [split_o[-1] + s for s in pc['points_split'][1:]]
    # the two preparation stages are bit-exact, so the model sees identical inputs
    assert split_o == split_d
    assert np.array_equal(np.concatenate(pts_o), det_info['points'][0].cpu().numpy())
    assert np.array_equal(np.concatenate(crops_o), torch.cat(crops_d).cpu().numpy())
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cfg = dict(fusion='C', affinity_op='multiply', softmax_mode='none', neg_threshold=0.2, score_arch='branch_cls')
    with torch.no_grad():
        o = R.tracking_forward(sd, cfg, torch.from_numpy(np.concatenate(crops_o)),
                               torch.from_numpy(np.concatenate(pts_o)).unsqueeze(0),
                               torch.tensor(split_o, dtype=torch.float32).unsqueeze(0), [5, 6])
    tm = model.test_mode
    assert (det_s - o[0][tm]).abs().max() < TOL
    assert (link_s[0] - o[1][0][tm:tm + 1]).abs().max() < TOL
    assert (new_s - o[2][tm]).abs().max() < TOL and (end_s - o[3][tm]).abs().max() < TOL


def test_sequence_pipeline_overlapped_equals_serial_equals_direct_calls():
    """mmmot_amd/pipeline.py: frame t+1's H2D + point gather + crop/resize on a side stream under pair t's forward
    returns BITWISE the scores of the serial order, and both equal the plain per-pair chain of the first test."""
    from mmmot_amd.crops import crop_resize_u8
    from mmmot_amd.pipeline import FrameFeed, SequencePipeline, stage_times
    from mmmot_amd.synth import make_frame
    S = 64
    model = TrackingNet(**KW)
    init_module(model, seed=0)
    model.eval().cuda()
    frames = [make_frame(90 + t, 20000, 4 + t % 3) for t in range(6)]
    feeds = [FrameFeed(*f) for f in frames]
    res_o = SequencePipeline(model, S, overlap=True).run(feeds)
    res_s = SequencePipeline(model, S, overlap=False).run(feeds)
    assert len(res_o) == len(res_s) == 5
    for a, b in zip(res_o, res_s):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1][0], b[1][0]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    # direct chain for pair (2, 3): the un-batched prep_points, u8 crops, one model call
    crops, pts, split, ds = [], [], [0], []
    for img, sweep, info, dets in frames[2:4]:
        crops.append(crop_resize_u8(torch.from_numpy(img).cuda(), dets['bbox'], S))
        pc = prep_points(torch.from_numpy(sweep).cuda(), info, dets, without_reflectivity=True)
        pts.append(pc['points'])
        split += [split[-1] + s for s in pc['points_split'][1:]]
        ds.append(torch.tensor([len(dets['rotation_y'])]))
    det_info = {'points': torch.cat(pts).unsqueeze(0), 'points_split': torch.tensor(split, dtype=torch.float32).unsqueeze(0).cuda()}
    with torch.no_grad():
        det, links, new, end, _ = model(torch.cat(crops), det_info, ds)
    d = scores_for_solver(det, links, new, end, model.test_mode)
    a = res_o[2]
    assert torch.equal(a[0], d[0]) and torch.equal(a[1][0], d[1][0]) and torch.equal(a[2], d[2]) and torch.equal(a[3], d[3])
    st = stage_times(model, feeds, S)
    assert set(st) >= {'h2d', 'prep_points', 'crop_resize', 'forward', 'sum_of_parts'} and st['forward'] > 0

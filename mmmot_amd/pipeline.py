"""One tracked sequence, frame by frame: the rows of SURVEY section 8 chained the way the reference's evaluation loop
chains them, with the next frame's upload and preparation running under the current pair's forward.

Reference chain per frame pair (t-1, t): ``TestSequence._generate_img_lidar`` (dataset/test_seq_dataset.py:176-246: per
frame ``get_pointcloud`` = ``read_and_prep_points`` and the crop / resize / normalise loop) -> ``input.cuda()``
(eval_seq.py:145-149) -> ``TrackingModule.predict`` (tracking_model.py:68-83: model forward, then ``ortools_solve`` on the
selected score rows).  The reference prepares BOTH frames of every pair on the host (each frame twice over a sequence)
and runs everything back to back.

Here, per frame t:   H2D (image, sweep)  ->  ``prep_points``  ->  ``crop_resize_u8``         [stage A: once per frame]
      per pair:      ``TrackingNet.forward`` on (A[t-1], A[t])  ->  ``scores_for_solver``    [stage B]
``overlap=True`` queues stage A of frame t+1 on a side stream BEFORE the host waits for the scores of pair (t-1, t), so
the upload, the gather and the resize run beside / under the forward.  Both orders launch the same kernels on the same
inputs: their outputs are bitwise equal (tests/test_pipeline_gpu.py).  The solver, ID bookkeeping and the ego-motion
alignment of the reference's dataset code stay on the host and are not part of this module (the synthetic sequence has
an identity ego motion; `FrameFeed(point_transform=...)` is the hook for the alignment of a frame's extracted points).  No CPU fallback: every stage is a C-ABI kernel sequence on the device.
"""
import time

import numpy as np
import torch

from .crops import crop_resize_u8
from .points import prep_points_batched
from .tracker_glue import scores_for_solver


class FrameFeed:
    """Host side of one frame: pinned staging copies of the image and the sweep (what a loader thread would hand over)."""

    def __init__(self, img, sweep, info, dets, point_transform=None):
        self.img = torch.from_numpy(np.ascontiguousarray(img)).pin_memory()
        self.sweep = torch.from_numpy(np.ascontiguousarray(sweep, dtype=np.float32)).pin_memory()
        self.info, self.dets = info, dets
        # optional: applied to the EXTRACTED points (device tensor [Q, 3|4]) of this frame - where the reference aligns the
        # second frame of a pair to the first one's coordinates (align_points, dataset/test_seq_dataset.py:199-210)
        self.point_transform = point_transform


class SequencePipeline:
    def __init__(self, model, size=224, overlap=True, without_reflectivity=True):
        self.model, self.size, self.overlap = model, int(size), bool(overlap)
        self.wo_refl = without_reflectivity
        self.dev = next(model.parameters()).device
        self.side = torch.cuda.Stream(self.dev) if overlap else None
        self.stage_events = None   # set to [] to record HIP events per stage (serial order only)

    # ---- stage A: one frame onto the device and through the two preparation kernels ---------------------------
    def _prepare(self, feed):
        ev = self.stage_events
        marks = []

        def mark():
            if ev is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                marks.append(e)
        mark()
        img = feed.img.to(self.dev, non_blocking=True)
        sweep = feed.sweep.to(self.dev, non_blocking=True)
        mark()
        # the image-frustum filter and the per-box gather in ONE launch sequence, one split read-back
        pc = prep_points_batched([sweep], [feed.info], [feed.dets], without_reflectivity=self.wo_refl)[0]
        pts = pc['points'] if feed.point_transform is None else feed.point_transform(pc['points']).contiguous()
        mark()
        crops = crop_resize_u8(img, feed.dets['bbox'], self.size)
        mark()
        if ev is not None:
            ev.append(('prep', marks))
        return {'crops': crops, 'points': pts, 'split': np.asarray(pc['points_split'], dtype=np.int64),
                'n': int(crops.shape[0]), 'ready': None}

    def prepare(self, feed):
        if not self.overlap:
            return self._prepare(feed)
        cur = torch.cuda.current_stream(self.dev)
        with torch.cuda.stream(self.side):
            a = self._prepare(feed)
            a['ready'] = torch.cuda.Event()
            a['ready'].record(self.side)
        for t in (a['crops'], a['points']):
            t.record_stream(cur)   # allocated on the side stream, consumed on the main one
        return a

    # ---- stage B: the pair forward + the packed hand-off -------------------------------------------------------
    def launch_pair(self, a, b):
        """queue TrackingNet.forward on frames (a, b); returns the device outputs (nothing waits on the host)"""
        cur = torch.cuda.current_stream(self.dev)
        for x in (a, b):
            if x['ready'] is not None:
                cur.wait_event(x['ready'])
        crops = torch.cat([a['crops'], b['crops']])
        points = torch.cat([a['points'], b['points']]).unsqueeze(0)
        split = np.concatenate([a['split'], a['split'][-1] + b['split'][1:]])
        # the split is on the host already (prep_points read it back): hand it over as a CPU tensor - no D2H in forward
        det_info = {'points': points, 'points_split': torch.from_numpy(split.astype(np.float32)).unsqueeze(0)}
        ev = self.stage_events
        if ev is not None:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        with torch.no_grad():
            out = self.model(crops, det_info, [torch.tensor([a['n']]), torch.tensor([b['n']])])
        if ev is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            ev.append(('forward', [e0, e1]))
        return out

    def hand_off(self, out):
        det, links, new, end, _ = out
        return scores_for_solver(det, links, new, end, self.model.test_mode)

    def run(self, feeds, on_scores=None):
        """All pairs (t-1, t) of the sequence.  Returns the list of host score tuples (det, [link], new, end) - what
        ``ortools_solve`` is called with; ``on_scores(t, scores)`` is where the host solver would run."""
        res = []
        prev = self.prepare(feeds[0])
        nxt = self.prepare(feeds[1]) if len(feeds) > 1 else None
        for t in range(1, len(feeds)):
            cur = nxt
            out = self.launch_pair(prev, cur)
            # stage A of the next frame is queued before the host blocks on this pair's scores
            nxt = self.prepare(feeds[t + 1]) if t + 1 < len(feeds) else None
            sc = self.hand_off(out)
            if on_scores is not None:
                on_scores(t, sc)
            res.append(sc)
            prev = cur
        return res


def time_sequence(model, feeds, size=224, overlap=True, warm=3):
    """frames/s of the chain over ``feeds`` (wall clock, synchronised on both sides; ``warm`` untimed leading pairs)."""
    pipe = SequencePipeline(model, size, overlap=overlap)
    pipe.run(feeds[:warm + 1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = pipe.run(feeds)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return (len(feeds) - 1) / dt, dt, res


def stage_times(model, feeds, size=224):
    """Device ms per stage of the SERIAL order (HIP events on the one stream): h2d, prep_points, crop_resize, forward;
    the hand-off copy is what is left of the wall time."""
    pipe = SequencePipeline(model, size, overlap=False)
    pipe.run(feeds[:3])
    pipe.stage_events = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.run(feeds)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    acc = {'h2d': 0.0, 'prep_points': 0.0, 'crop_resize': 0.0, 'forward': 0.0}
    nprep = nfwd = 0
    for kind, m in pipe.stage_events:
        if kind == 'prep':
            acc['h2d'] += m[0].elapsed_time(m[1])
            acc['prep_points'] += m[1].elapsed_time(m[2])
            acc['crop_resize'] += m[2].elapsed_time(m[3])
            nprep += 1
        else:
            acc['forward'] += m[0].elapsed_time(m[1])
            nfwd += 1
    out = {'h2d': acc['h2d'] / nprep, 'prep_points': acc['prep_points'] / nprep, 'crop_resize': acc['crop_resize'] / nprep,
           'forward': acc['forward'] / nfwd}
    out['sum_of_parts'] = sum(out.values())
    out['wall_per_frame_serial'] = wall / nfwd
    return {k: round(v, 4) for k, v in out.items()}

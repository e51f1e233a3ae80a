"""PyTorch-ROCm custom operators of the HIP path (``torch.ops.mmmot.*``).

``TrackingNet.forward`` reaches libmmmot_hip.so through ONE registered operator per fused group, dispatched on the
CUDA (= HIP on ROCm) key only:

    mmmot::forward_batch(Tensor? crops, Tensor? points, int engine, int plan) -> Tensor[]
        the whole eval-mode frame-pair forward of reference modules/tracking_net.py:165-193 for a batch of
        samples: VGG trunk + SkipPool, PointNet, fusion, w_det, affinity + new/end, softmax.
        Returns [det (nR, Lt), link (flat), new (nR, Lt), end (nR, Lt)].
    mmmot::appearance(Tensor crops, int engine, int plan) -> Tensor      L x 512 image features (appear_net.py:178-190)
    mmmot::pointnet(Tensor points, int engine, int plan) -> Tensor       L x 512 LiDAR features (point_net.py:25-44)

``engine`` / ``plan`` are handles into this module's registries: the packed weights + workspace arena (an
``Engine``) and the integer tile / segment tables of one batch shape (a ``BatchPlan``) are host objects that hold
device memory; they are not tensors and do not belong in an operator signature.  There is no CPU kernel: CPU tensors
raise ``NotImplementedError`` from the dispatcher (no fallback).  Every operator has a Meta kernel (output shapes
from the plan), so the path can be traced with FakeTensors / ``torch.compile`` graphs can carry it as an opaque node.
The launches go to torch's current HIP stream; nothing synchronises.

Contract the schema cannot express (the handles hide mutable state):
  * every returned tensor is freshly allocated and no two outputs share storage (the engine's new / end scores live in
    one buffer: ``end`` is copied out) - what the functional custom-op contract and the Meta kernels promise;
  * the engine's workspace arena is scratch shared by all calls on that engine.  ``Engine.forward`` refuses a second
    thread while a forward is being issued and orders a forward behind the previous one when the stream changed, so
    eager use from any stream is safe; what is NOT supported is two calls on the SAME engine issued concurrently or
    re-ordered as if independent (a compiler that treats the op as pure may do the latter only for calls whose
    results are unused) - use one TrackingNet per concurrent stream.
"""
import itertools
import weakref

import torch

_LIB = torch.library.Library('mmmot', 'DEF')
_ENGINES = weakref.WeakValueDictionary()
_PLANS = weakref.WeakValueDictionary()
_ids = itertools.count(1)


def handle(obj, registry):
    """Stable integer handle of an Engine / BatchPlan (kept on the object; the registry holds a weak reference)."""
    h = getattr(obj, '_mmmot_handle', None)
    if h is None:
        h = next(_ids)
        obj._mmmot_handle = h
    registry[h] = obj
    return h


def engine_handle(engine):
    return handle(engine, _ENGINES)


def plan_handle(plan):
    return handle(plan, _PLANS)


def _lookup(engine, plan):
    try:
        return _ENGINES[engine], _PLANS[plan]
    except KeyError:
        raise RuntimeError('mmmot: stale engine / plan handle (the owning TrackingNet or BatchPlan was released)')


def _forward_batch(crops, points, engine, plan):
    eng, pl = _lookup(engine, plan)
    out = eng.forward(pl, crops, points)
    return [out['det'], out['link'], out['new'], out['end'].clone()]  # new / end share one buffer in the engine


def _forward_batch_meta(crops, points, engine, plan):
    _, pl = _lookup(engine, plan)
    ref = crops if crops is not None else points
    n_link = pl.pair_tiles.R
    mk = lambda *s: ref.new_empty(s, dtype=torch.float32)
    return [mk(pl.nR, pl.Lt), mk(n_link), mk(pl.nR, pl.Lt), mk(pl.nR, pl.Lt)]


def _appearance(crops, engine, plan):
    eng, pl = _lookup(engine, plan)
    eng.dev = crops.device
    cat = eng.buf('cat', pl.Lt, 1024)
    eng.appearance(pl, crops, cat)
    return cat[:, :512].clone()


def _pointnet(points, engine, plan):
    eng, pl = _lookup(engine, plan)
    eng.dev = points.device
    cat = eng.buf('cat', pl.Lt, 1024)
    eng.pointnet(pl, points, cat)
    return cat[:, 512:].clone()


def _feat_meta(x, engine, plan):
    _, pl = _lookup(engine, plan)
    return x.new_empty((pl.Lt, 512), dtype=torch.float32)


_LIB.define('forward_batch(Tensor? crops, Tensor? points, int engine, int plan) -> Tensor[]')
_LIB.impl('forward_batch', _forward_batch, 'CUDA')
_LIB.impl('forward_batch', _forward_batch_meta, 'Meta')
_LIB.define('appearance(Tensor crops, int engine, int plan) -> Tensor')
_LIB.impl('appearance', _appearance, 'CUDA')
_LIB.impl('appearance', _feat_meta, 'Meta')
_LIB.define('pointnet(Tensor points, int engine, int plan) -> Tensor')
_LIB.impl('pointnet', _pointnet, 'CUDA')
_LIB.impl('pointnet', _feat_meta, 'Meta')

#!/usr/bin/env python
"""Which call sites launch mmmot_segment_mean during a whole-network training step, and with what shapes (segments,
channels, longest segment): the strided-sum kernel serves a dozen different reductions of the backward pass, and its
launches only show up as one name in a kernel trace.  GPU box only.

    python tools/train_segment_calls.py | grep "calls  nseg"
"""
import sys, os, traceback, collections, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
sys.argv = [sys.argv[0], '--whole']
from mmmot_amd import ops as O
calls = collections.OrderedDict()
orig = O.HipOps.segment_mean
def wrapped(self, X, C, segs, out, *a, **k):
    st = traceback.extract_stack(limit=5)
    site = ' < '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in reversed(st[:-1]))
    cnt = segs.h_count if hasattr(segs, 'h_count') else None
    key = (site, segs.n, C, int(cnt.max()) if cnt is not None else -1, int(cnt.sum()) if cnt is not None else -1)
    calls[key] = calls.get(key, 0) + 1
    return orig(self, X, C, segs, out, *a, **k)
O.HipOps.segment_mean = wrapped
import runpy
try:
    runpy.run_path(os.path.join(os.path.dirname(__file__), 'profile_train_step.py'), run_name='__main__')
except SystemExit:
    pass
for k, v in sorted(calls.items(), key=lambda kv: -kv[0][3] * kv[1]):
    print('%4d calls  nseg %5d  C %5d  max rows %7d  total rows %8d   %s' % (v, k[1], k[2], k[3], k[4], k[0]))

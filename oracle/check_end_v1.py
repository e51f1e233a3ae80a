#!/usr/bin/env python
"""Evidence for DESIGN.md "out of scope": the reference's NewEndIndicator_v1 (modules/new_end.py:5-40, selected by
end_arch='v1', modules/tracking_net.py:65-70) cannot run through the reference's own eval forward.

Its w_new / w_end are reshaped with .view((M, -1)) / .view((N, -1)) (new_end.py:37-38), i.e. [M][rows] / [N][rows], and
TrackingNet.forward concatenates them with the [rows][N] / [rows][M] zero padding along dim 1 (tracking_net.py:183-189):
the call raises unless N == M == rows (= 3).  Run in the build container (needs /root/reference; test infrastructure,
never imported by the product):

    python oracle/check_end_v1.py
"""
import contextlib
import io
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # noqa: E402
from mmmot_amd.synth import make_pair  # noqa: E402
from mmmot_amd.weights import generate_state_dict  # noqa: E402


def main():
    rm = G.import_reference()
    for fusion in 'ABC':
        kw = dict(G.BASE, score_fusion_arch=fusion, affinity_op='multiply', softmax_mode='none', seq_len=2,
                  end_arch='v1')
        with contextlib.redirect_stdout(io.StringIO()):
            m = rm.TrackingNet(**kw)
        m.load_state_dict(generate_state_dict(m.state_dict(), seed=0))
        m.eval()
        for N, M in ((3, 3), (5, 7), (1, 1), (12, 12)):
            dets, info, split = make_pair(N, M, 32, 20, 7, ragged=True)
            try:
                with torch.no_grad():
                    out = m(dets, info, split)
                print('fusion %s N=%d M=%d: runs, new %s end %s' % (fusion, N, M, tuple(out[2].shape), tuple(out[3].shape)))
            except RuntimeError as e:
                print('fusion %s N=%d M=%d: RuntimeError: %s' % (fusion, N, M, str(e).splitlines()[0][:120]))


if __name__ == '__main__':
    main()

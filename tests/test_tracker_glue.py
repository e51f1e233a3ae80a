"""Selection + packed host copy for the reference solver (reference tracking_model.py:72-75)."""
import torch

from mmmot_amd.tracker_glue import scores_for_solver


def test_selection_matches_reference_indexing():
    g = torch.Generator().manual_seed(3)
    N, M = 4, 6
    det, new, end = (torch.rand(3, N + M, generator=g) for _ in range(3))
    link = [torch.rand(3, N, M, generator=g)]
    for tm in (0, 1, 2):
        d, l, n, e = scores_for_solver(det, link, new, end, tm)
        assert torch.equal(d, det[tm]) and torch.equal(n, new[tm]) and torch.equal(e, end[tm])
        assert len(l) == 1 and l[0].shape == (1, N, M) and torch.equal(l[0], link[0][tm:tm + 1])
        assert d.device.type == 'cpu'

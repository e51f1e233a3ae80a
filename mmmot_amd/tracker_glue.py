"""Hand-off from the device forward to the reference's host-side tracker.

``TrackingModule.predict`` (reference tracking_model.py:68-83) indexes the network outputs by
``test_mode`` and passes them to ``ortools_solve``, which reads every score with ``.item()``
(solvers.py:32-45): O(N*M) device synchronisations once the scores live on the GPU.  ``scores_for_solver``
does the same selection and moves the selected rows to the host in ONE packed copy, returning CPU tensors
with exactly the shapes the solver indexes.  The solver itself stays the reference's.
"""
import torch


def scores_for_solver(det_score, link_scores, new_score, end_score, test_mode):
    """(det 3xL, [link 3xNxM ...], new 3xL, end 3xL) -> CPU (det L, [link 1xNxM ...], new L, end L)
    as consumed at reference tracking_model.py:72-75."""
    tm = int(test_mode)
    parts = [det_score[tm].reshape(-1), new_score[tm].reshape(-1), end_score[tm].reshape(-1)]
    parts += [l[tm:tm + 1].reshape(-1) for l in link_scores]
    flat = torch.cat(parts).to('cpu')  # one device-to-host transfer
    L = det_score.shape[1]
    det, new, end = flat[0:L], flat[L:2 * L], flat[2 * L:3 * L]
    links, o = [], 3 * L
    for l in link_scores:
        n = l.shape[1] * l.shape[2]
        links.append(flat[o:o + n].view(1, l.shape[1], l.shape[2]))
        o += n
    return det, links, new, end


def predict_scores(model, det_imgs, det_info, det_split):
    """Mirror of the first half of ``TrackingModule.predict``: forward + selection, ready for
    ``ortools_solve(det, links, new, end, det_split)``."""
    with torch.no_grad():
        det_score, link_score, new_score, end_score, _ = model(det_imgs, det_info, det_split)
    return scores_for_solver(det_score, link_score, new_score, end_score, model.test_mode)

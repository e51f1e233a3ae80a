"""Property tests of the integer tables of a BatchPlan (mmmot_amd/plan.py) on random ragged batches: the tables are
what every kernel indexes with, so their invariants are checked exhaustively on the host (hypothesis, no GPU)."""
import numpy as np
from hypothesis import given, settings, strategies as st

from mmmot_amd.plan import BatchPlan, RowTiles


@st.composite
def batches(draw):
    B = draw(st.integers(1, 4))
    samples = []
    for _ in range(B):
        nf = draw(st.integers(2, 3))
        fc = [draw(st.integers(1, 9)) for _ in range(nf)]
        cnt = [draw(st.sampled_from([1, 1, 2, 7, 130, 300])) for _ in range(sum(fc))]
        samples.append((fc, np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)))
    return samples


def check_tiles(t, counts):
    assert t.R == sum(counts) and t.G == len(counts)
    rows = np.concatenate([np.arange(r0, r0 + n) for r0, n in zip(t.h_row0, t.h_nrows)]) if t.T else np.zeros(0)
    assert np.array_equal(rows, np.arange(t.R)), 'tiles must partition the rows in order'
    assert (t.h_nrows >= 1).all() and (t.h_nrows <= 128).all()
    for g, c in enumerate(counts):
        sel = t.h_group == g
        assert t.h_nrows[sel].sum() == c
        idx = np.nonzero(sel)[0]
        assert idx[0] == t.h_g_tile0[g] and len(idx) == t.h_g_ntiles[g] and (np.diff(idx) == 1).all()
        assert t.h_row0[idx[0]] == t.h_g_row0[g]


@settings(max_examples=40, deadline=None)
@given(batches())
def test_plan_tables_are_consistent(samples):
    p = BatchPlan(samples, 32, 'cpu')
    L = [sum(fc) for fc, _ in samples]
    assert p.Lt == sum(L) and p.P == sum(int(ps[-1]) for _, ps in samples)
    # device views of the single upload equal the host tables
    for obj, names in ((p.pt_tiles, ('row0', 'nrows', 'group', 'g_tile0', 'g_ntiles', 'g_count')),
                       (p.v_segs, ('start', 'count', 'stride', 'group'))):
        for n in names:
            assert np.array_equal(getattr(obj, n).numpy(), getattr(obj, 'h_' + n).astype(np.int32))
    check_tiles(p.pt_tiles, p.P_b)
    check_tiles(p.det_tiles, L)
    # detection-aligned point tiles: no tile straddles two detections
    cnts = np.diff(p.pt_split)
    det_of_row = np.repeat(np.arange(p.Lt), cnts)
    for r0, n in zip(p.ptd_tiles.h_row0, p.ptd_tiles.h_nrows):
        assert det_of_row[r0] == det_of_row[r0 + n - 1]
    assert np.array_equal(p.row_det.numpy(), det_of_row)
    assert np.array_equal(p.tile_det.numpy(), det_of_row[p.ptd_tiles.h_row0])
    # pair groups: one per (association pair, modality row); rows = N*M; the flat link offsets follow them
    assert len(p.pairs) == sum(len(fc) - 1 for fc, _ in samples)
    g = 0
    for pi, (b, a0, N, b0, M) in enumerate(p.pairs):
        assert p.link_off[pi] == p.pair_tiles.h_g_row0[g]
        for ri in range(3):
            assert p.h_pg_N[g] == N and p.h_pg_M[g] == M and p.pair_tiles.h_g_count[g] == N * M
            assert p.h_pg_aoff[g] == ri * p.Lt + a0 and p.h_pg_boff[g] == ri * p.Lt + b0
            g += 1
    # new / end scatter map: every (score kind, row, detection) slot is written at most once; "new" never lands on a
    # first-frame detection, "end" never on a last-frame one (the eval-mode zero padding, tracking_net.py:183-189)
    om = p.v_omap.numpy()
    assert len(np.unique(om)) == len(om) and om.min() >= 0 and om.max() < 2 * 3 * p.Lt
    kind, rem = om // (3 * p.Lt), om % (3 * p.Lt)
    det = rem % p.Lt
    first, last = np.zeros(p.Lt, bool), np.zeros(p.Lt, bool)
    for b, (fc, _) in enumerate(samples):
        d0 = int(p.det_off[b])
        first[d0:d0 + fc[0]] = True
        last[d0 + sum(fc[:-1]):d0 + sum(fc)] = True
    assert not first[det[kind == 0]].any() and not last[det[kind == 1]].any()
    # strided segments of the new / end means stay inside their pair group
    vs = p.v_segs
    for s in range(vs.n):
        g = int(vs.h_group[s])
        lo, hi = int(p.pair_tiles.h_g_row0[g]), int(p.pair_tiles.h_g_row0[g]) + int(p.pair_tiles.h_g_count[g])
        assert lo <= vs.h_start[s] and vs.h_start[s] + (vs.h_count[s] - 1) * vs.h_stride[s] < hi


@settings(max_examples=60, deadline=None)
@given(st.lists(st.integers(1, 700), min_size=1, max_size=6), st.sampled_from([128, 4096]))
def test_row_tiles_partition_rows(counts, tile):
    t = RowTiles(counts, 'cpu', tile=tile)
    rows = np.concatenate([np.arange(r0, r0 + n) for r0, n in zip(t.h_row0, t.h_nrows)])
    assert np.array_equal(rows, np.arange(sum(counts))) and (t.h_nrows <= tile).all()
    assert [int(t.h_nrows[t.h_group == g].sum()) for g in range(len(counts))] == counts

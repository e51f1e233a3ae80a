"""Batch plan: every integer table the device kernels need for one batch of
frame samples, built once on the host (numpy) and uploaded.

A *sample* is what the reference calls one dataset item: the detections of
``sample_max_len`` consecutive frames (always 2 in the shipped configs,
reference experiments/*/config.yaml:23) with their crops and LiDAR points.
The reference processes exactly one sample per forward (batch_size 1,
eval_seq.py:153); the plan generalises to B independent samples per launch:
normalisation statistics never cross samples (SURVEY 8a GN note), so each
sample is a *group* in the row-tile tables.

Row spaces (all "position-major", channels contiguous):
  points  P_t rows   group = sample                 (PointNet shared MLP)
  dets    L_t rows   group = sample                 (fusion, PointNet head)
  F       nR*L_t     no groups                      (w_det; nR modality rows)
  pairs   sum nR*N*M group = (assoc pair, row)      (affinity N x M blocks)
  V       sum nR*(M+N) group = (pair, row, new|end) (new/end heads)
"""
import numpy as np
import torch

TILE = 128


class TableUpload:
    """Collects the int32 host tables of a plan and moves them to the device in ONE copy.

    A plan holds ~100 small integer tables; uploading each with its own ``.to(device)`` costs ~35 us of host time
    apiece (3.6 ms per reference-shaped call, measured with tools/profile_python_overhead.py) - more than the device
    needs for the whole forward at B=1.  Pass a ``TableUpload`` instead of a device to ``RowTiles`` / ``Segments``;
    ``flush()`` concatenates everything (16-byte aligned pieces), does one host-to-device copy and hands every owner
    a view of the device buffer."""

    def __init__(self, device):
        self.device = device
        self.items = []

    def add(self, obj, attr, arr):
        self.items.append((obj, attr, np.ascontiguousarray(arr, dtype=np.int32).reshape(-1)))

    def flush(self):
        if not self.items:
            return
        offs, total = [], 0
        for _, _, a in self.items:
            offs.append(total)
            total += (a.size + 3) // 4 * 4
        buf = np.zeros(max(total, 4), np.int32)
        for o, (_, _, a) in zip(offs, self.items):
            buf[o:o + a.size] = a
        dev = torch.from_numpy(buf).to(self.device)
        for o, (obj, attr, a) in zip(offs, self.items):
            setattr(obj, attr, dev[o:o + a.size])
        self.items = []


def _put(obj, device, **tables):
    """device tensors for the named int32 tables: immediately, or through a TableUpload"""
    for attr, arr in tables.items():
        if isinstance(device, TableUpload):
            device.add(obj, attr, arr)
        else:
            setattr(obj, attr, torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int32)).to(device))


class RowTiles:
    """Tiles of <=128 rows over contiguous groups of rows.

    ``sub_counts`` (optional, one list per group, summing to the group's count): the tiles additionally never
    straddle these sub-segments (detections inside a sample) - the layout the fused "normalise + ReLU +
    per-detection mean" GEMM epilogue needs.  Such tilings are ``ragged``: partial tiles occur inside a group
    and the finalize kernel is handed the per-tile row counts."""

    def __init__(self, counts, device, sub_counts=None, tile=TILE):
        counts = [int(c) for c in counts]
        # rows per tile (128 for the GEMM kernels; up to 4096 for the Gram super-tiles), optionally one value per group
        tiles_g = [int(t) for t in tile] if isinstance(tile, (list, tuple, np.ndarray)) else [int(tile)] * len(counts)
        row0, nrows, group, g_tile0, g_ntiles, g_row0 = [], [], [], [], [], []
        sub_tile0, sub_ntiles = [], []
        r = 0
        for g, c in enumerate(counts):
            TILE = tiles_g[g]
            g_tile0.append(len(row0))
            g_row0.append(r)
            subs = [c] if sub_counts is None else [int(x) for x in sub_counts[g]]
            if sum(subs) != c:
                raise ValueError('sub-segment counts of group %d do not add up' % g)
            for sc in subs:
                sub_tile0.append(len(row0))
                nt = (sc + TILE - 1) // TILE
                for t in range(nt):
                    row0.append(r + t * TILE)
                    nrows.append(min(TILE, sc - t * TILE))
                    group.append(g)
                sub_ntiles.append(nt)
                r += sc
            g_ntiles.append(len(row0) - g_tile0[-1])
        self.ragged = sub_counts is not None
        self.R = r
        self.T = len(row0)
        self.G = len(counts)
        self.h_row0 = np.asarray(row0, np.int32)
        self.h_nrows = np.asarray(nrows, np.int32)
        self.h_group = np.asarray(group, np.int32)
        self.h_g_tile0 = np.asarray(g_tile0, np.int32)
        self.h_g_ntiles = np.asarray(g_ntiles, np.int32)
        self.h_g_count = np.asarray(counts, np.int32)
        self.h_g_row0 = np.asarray(g_row0, np.int32)
        self.h_sub_tile0 = np.asarray(sub_tile0, np.int32)
        self.h_sub_ntiles = np.asarray(sub_ntiles, np.int32)
        _put(self, device, row0=self.h_row0, nrows=self.h_nrows, group=self.h_group, g_tile0=self.h_g_tile0,
             g_ntiles=self.h_g_ntiles, g_count=self.h_g_count, g_row0=self.h_g_row0)


class HalfTiles:
    """The 64-row halves of a RowTiles tiling as a statistics tiling: entry 2t+h, possibly empty."""

    def __init__(self, tiles, device):
        n = tiles.h_nrows.astype(np.int64)
        self.h_nrows = np.stack([np.minimum(n, 64), np.clip(n - 64, 0, 64)], 1).reshape(-1).astype(np.int32)
        self.T, self.G, self.R, self.ragged = 2 * tiles.T, tiles.G, tiles.R, True
        self.h_g_tile0, self.h_g_ntiles, self.h_g_count = 2 * tiles.h_g_tile0, 2 * tiles.h_g_ntiles, tiles.h_g_count
        _put(self, device, nrows=self.h_nrows, g_tile0=self.h_g_tile0, g_ntiles=self.h_g_ntiles, g_count=self.h_g_count)


class Segments:
    def __init__(self, start, count, stride, group, device, div=None):
        self.h_div = None if div is None else np.asarray(div, np.int64)
        self.div = None  # divisor per segment (default: its row count)
        self.n = len(start)
        self.h_start, self.h_count = np.asarray(start, np.int64), np.asarray(count, np.int64)
        self.h_stride, self.h_group = np.asarray(stride, np.int64), np.asarray(group, np.int64)
        _put(self, device, start=self.h_start, count=self.h_count, stride=self.h_stride, group=self.h_group)
        if div is not None:
            _put(self, device, div=self.h_div)


class BatchPlan:
    """samples: list of (frame_counts, points_split) with frame_counts a list of
    per-frame detection counts and points_split the sample-local cumulative point
    offsets (L_b + 1 integers, first 0) - the reference's ``dets_split`` /
    ``det_info['points_split']`` (dataset/test_seq_dataset.py:224-244)."""

    def __init__(self, samples, crop_hw, device, rows=(0, 1, 2), use_points=True, use_images=True):
        self.device = device
        device = up_all = TableUpload(device)  # every table below goes to the device in one copy (flush at the end)
        self.rows = tuple(rows)
        self.nR = nR = len(self.rows)
        self.S = int(crop_hw)
        self.B = len(samples)
        frame_counts = []
        L = []
        for fc, _ in samples:
            fc = [int(x) for x in fc]
            if len(fc) < 2 or min(fc) < 1:
                raise ValueError('every sample needs >= 2 frames with >= 1 detection each, got %r' % (fc,))
            frame_counts.append(fc)
            L.append(sum(fc))
        self.frame_counts = frame_counts
        self.L = L
        self.det_off = np.concatenate([[0], np.cumsum(L)]).astype(np.int64)
        self.Lt = Lt = int(self.det_off[-1])
        self._tbl = {}

        def up(a, name=None):  # named int32 table of the plan itself -> attribute set by the flush
            name = name or '_t%d' % len(self._tbl)
            self._tbl[name] = True
            up_all.add(self, name, a)
            return name

        # ---- point space -------------------------------------------------
        self.use_points = use_points
        if use_points:
            gsplit = [0]
            P_b = []
            for b, (_, ps) in enumerate(samples):
                ps = np.asarray(ps).astype(np.int64).reshape(-1)
                if ps.shape[0] != L[b] + 1 or ps[0] != 0:
                    raise ValueError('points_split of sample %d must have L+1=%d entries starting at 0' % (b, L[b] + 1))
                cnt = np.diff(ps)
                if (cnt < 1).any():
                    raise ValueError('every detection needs >= 1 point (reference pads empty boxes with one '
                                     'zero point, point_cloud/preprocess.py:80-81)')
                base = gsplit[-1]
                gsplit.extend((base + ps[1:]).tolist())
                P_b.append(int(ps[-1]))
            self.pt_split = np.asarray(gsplit, np.int64)
            self.P = int(self.pt_split[-1])
            if self.P >= 2 ** 31 - TILE:
                raise ValueError('too many points for int32 row indices')
            self.P_b = P_b
            self.pt_tiles = RowTiles(P_b, device)
            cnts = np.diff(self.pt_split)
            up(np.repeat(np.arange(Lt, dtype=np.int64), cnts), 'row_det')
            det_sample = np.repeat(np.arange(self.B), L)
            self.det_segs = Segments(self.pt_split[:-1], cnts, np.ones(Lt), det_sample, device)
            # detection-aligned point tiles + "tiles of one detection" segments (fused PointNet epilogues:
            # the GEMM writes per-tile column sums, the segment mean adds a detection's tiles and divides
            # by its point count)
            per_sample = [cnts[self.det_off[b]:self.det_off[b + 1]] for b in range(self.B)]
            self.ptd_tiles = RowTiles(P_b, device, sub_counts=per_sample)
            self.det_tile_segs = Segments(self.ptd_tiles.h_sub_tile0, self.ptd_tiles.h_sub_ntiles, np.ones(Lt),
                                          det_sample, device, div=cnts)
            # super-tiles of <= 4096 rows inside a sample: second-moment (Gram) statistics of a layer's input.  One
            # workgroup walks one super-tile, so small point clouds (a single reference-shaped pair: ~7 000 points)
            # get shorter super-tiles - 4096-row ones would leave two workgroups to do the whole pass (0.21 ms at
            # B = 1, profiles/README.md); large batches keep 4096 (fewer float64 partials to merge)
            # (chosen per SAMPLE, from its own point count: a sample's statistics do not depend on what it is batched
            # with - tests/test_parity_gpu.py::test_batched_equals_single_and_is_deterministic)
            gt = [min(4096, max(128, 128 * -(-p_b // (32 * 128)))) for p_b in P_b]
            self.gram_tiles = RowTiles(P_b, device, tile=gt)
            # the same for PointNet_v1.conv1 (64-channel input, gathered per-detection bias): super-tiles of <= 2048 rows
            # cut along DETECTIONS, so that a super-tile's column sums belong to one bias row (mmmot_gn_finalize_gram_dbias)
            gt64 = [min(2048, max(128, 128 * -(-p_b // (32 * 128)))) for p_b in P_b]
            self.gram64_tiles = RowTiles(P_b, device, sub_counts=per_sample, tile=gt64)
            up(np.repeat(np.arange(Lt, dtype=np.int64), self.gram64_tiles.h_sub_ntiles), 'gram64_tile_det')
            # 64-row half tiles of ptd_tiles (the A-resident GEMM emits its partials per wave = per half tile)
            self.ptd_half = HalfTiles(self.ptd_tiles, device)
            self.det_half_segs = Segments(2 * self.ptd_tiles.h_sub_tile0, 2 * self.ptd_tiles.h_sub_ntiles,
                                          np.ones(Lt), det_sample, device, div=cnts)
            # tile t belongs to detection tile_det[t]
            up(np.repeat(np.arange(Lt, dtype=np.int64), self.ptd_tiles.h_sub_ntiles), 'tile_det')

        # ---- detection spaces -------------------------------------------
        self.det_tiles = RowTiles(L, device)
        self.F_tiles = RowTiles([nR * Lt], device)
        self.crop_segs = {}  # (h*w) -> Segments over crops, built lazily

        # ---- association pairs ------------------------------------------
        pairs = []  # (sample, a_det0, N, b_det0, M)
        for b, fc in enumerate(frame_counts):
            d0 = int(self.det_off[b])
            for f in range(len(fc) - 1):
                pairs.append((b, d0, fc[f], d0 + fc[f], fc[f + 1]))
                d0 += fc[f]
        self.pairs = pairs
        g_N, g_M, g_aoff, g_boff, g_cnt = [], [], [], [], []
        for (b, a0, N, b0, M) in pairs:
            for ri in range(nR):
                g_N.append(N); g_M.append(M)
                g_aoff.append(ri * Lt + a0); g_boff.append(ri * Lt + b0)
                g_cnt.append(N * M)
        self.pair_tiles = RowTiles(g_cnt, device)
        if self.pair_tiles.R >= 2 ** 31 - TILE:
            raise ValueError('too many detection pairs for int32 row indices')
        up(g_N, 'pg_N'); up(g_M, 'pg_M'); up(g_aoff, 'pg_aoff'); up(g_boff, 'pg_boff')
        self.h_pg_N, self.h_pg_M = np.asarray(g_N), np.asarray(g_M)
        self.h_pg_aoff, self.h_pg_boff = np.asarray(g_aoff), np.asarray(g_boff)
        self.max_nm = int(max(n + m for n, m in zip(g_N, g_M)))
        # mmmot_gemm_args.pair_uniform32: every 32-row block of every pair tile lies inside one previous-frame index i -
        # M % 32 == 0 in every group AND every tile starts a multiple of 32 rows into its group (RowTiles cuts a group
        # every TILE = 128 rows, so the second condition holds by construction; checked, not assumed - ADVICE r4)
        PT = self.pair_tiles
        self.pair_uniform32 = bool(len(g_M)) and bool((self.h_pg_M % 32 == 0).all()) and \
            bool(((PT.h_row0 - PT.h_g_row0[PT.h_group]) % 32 == 0).all())
        self.link_off = [int(self.pair_tiles.h_g_row0[p * nR]) for p in range(len(pairs))]

        # ---- new / end vectors ------------------------------------------
        v_cnt, s_start, s_count, s_stride, s_group, omap = [], [], [], [], [], []
        for p, (b, a0, N, b0, M) in enumerate(pairs):
            for ri in range(nR):
                g = p * nR + ri
                r0 = int(self.pair_tiles.h_g_row0[g])
                # "new": mean over previous-frame index i, one row per current detection j
                v_cnt.append(M)
                for j in range(M):
                    s_start.append(r0 + j); s_count.append(N); s_stride.append(M); s_group.append(g)
                    omap.append(0 * nR * Lt + ri * Lt + b0 + j)
                # "end": mean over current-frame index j, one row per previous detection i
                v_cnt.append(N)
                for i in range(N):
                    s_start.append(r0 + i * M); s_count.append(M); s_stride.append(1); s_group.append(g)
                    omap.append(1 * nR * Lt + ri * Lt + a0 + i)
        self.v_tiles = RowTiles(v_cnt, device)
        self.v_segs = Segments(s_start, s_count, s_stride, s_group, device)
        up(omap, 'v_omap')
        # global-average-pool segments of the four VGG stages (S/4 .. S/32 maps): built now, with everything else
        if use_images and self.S >= 32:
            for sh in (4, 8, 16, 32):
                self._crop_segments((self.S // sh) ** 2, device)
        up_all.flush()

    def _crop_segments(self, hw, device):
        """Global average pool of every crop's hw pixels.  One workgroup reduces one segment, so with few crops (a single
        reference-shaped pair: 22 crops x 3136 pixels at stage 0) a segment per crop leaves 22 workgroups walking 3136
        rows each - 0.3 ms at B = 1.  Maps of more than 256 pixels are therefore pooled in two levels: 256-row chunks of a crop are summed (divisor 1),
        a second pass adds a crop's S partial rows and divides by hw.  Returns (first, second | None, rows of partials)."""
        Lt = self.Lt
        S = -(-hw // 256)  # fixed 256-row chunks: the summation order of a crop does not depend on the batch it is in
        if S <= 1:
            self.crop_segs[hw] = (Segments(np.arange(Lt) * hw, np.full(Lt, hw), np.ones(Lt), np.zeros(Lt), device), None, 0)
            return self.crop_segs[hw]
        chunk = 256
        l, c = np.repeat(np.arange(Lt), S), np.tile(np.arange(S), Lt)
        start = l * hw + c * chunk
        count = np.minimum(chunk, hw - c * chunk)
        first = Segments(start, count, np.ones(Lt * S), np.zeros(Lt * S), device, div=np.ones(Lt * S))
        second = Segments(np.arange(Lt) * S, np.full(Lt, S), np.ones(Lt), np.zeros(Lt), device, div=np.full(Lt, hw))
        self.crop_segs[hw] = (first, second, Lt * S)
        return self.crop_segs[hw]

    def crop_segments(self, hw):
        """see _crop_segments"""
        if hw not in self.crop_segs:
            self._crop_segments(hw, self.device)
        return self.crop_segs[hw]

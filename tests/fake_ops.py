"""TEST-ONLY torch emulation of the C-ABI operator semantics (include/mmmot_hip.h).

Purpose: the GPU-less (``-m "not gpu"``) suite uses it to validate the HOST
logic - weight packing / BN + STN folding, the batch plan's integer tables and
the engine's launch schedule - against the golden vectors, so that a failure
on the GPU box isolates to the HIP kernels.  It lives under tests/, is injected
explicitly with ``TrackingNet.set_ops`` and is unreachable from the product
path (``mmmot_amd`` never imports tests; HipOps rejects CPU tensors).

Each method documents the contract of one entry point in executable form; the
per-kernel GPU tests compare the HIP kernels against these same functions.
"""
import torch

ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2


def _act(v, act):
    if act == ACT_RELU:
        return torch.relu(v)
    if act == ACT_SIGMOID:
        return torch.sigmoid(v)
    return v


def _rows_groups(tiles):
    """row -> group for every row covered by the tile table."""
    g = torch.empty(tiles.R, dtype=torch.long)
    for t in range(tiles.T):
        r0, n = int(tiles.h_row0[t]), int(tiles.h_nrows[t])
        g[r0:r0 + n] = int(tiles.h_group[t])
    return g


class TorchOps:
    name = 'torch-emulation'

    def __init__(self, dtype=torch.float32):
        self.dtype = dtype
        self.range = [0, 0, 0, 0]  # emulation of the trunk range-guard counters (mmmot_trunk_range_read)
        self.range_bound = None    # the caller's counter block (mmmot_trunk_range_bind): int32 [4] tensor

    @staticmethod
    def _osv(oscale, Cout):
        """per-output-channel scale [Cout] (a python float = the same scale for every channel) -> [1, Cout, 1, 1]"""
        if torch.is_tensor(oscale):
            return oscale.reshape(1, Cout, 1, 1).double()
        return torch.full((1, Cout, 1, 1), float(oscale), dtype=torch.float64)

    def _count(self, i, n):
        if self.range_bound is not None:
            self.range_bound[i] += int(n)
        else:
            self.range[i] += int(n)

    def _guard(self, rows, q8):
        """rows: post-ReLU activations about to be stored; counts what the epilogue's range guard counts"""
        if q8:
            self._count(0, (rows > 1792.0).sum())
        self._count(1, (rows > 65000.0).sum())

    def trunk_range_bind(self, counters):
        self.range_bound = counters

    def trunk_range_read(self, device=None, reset=True):
        r = tuple(self.range)
        if reset:
            self.range = [0, 0, 0, 0]
        return r

    def conv3x3(self, inp, wp, bias, out, L, H, W, Cin, Cout, first, pool):
        if first:
            x = inp.view(L, 3, H, W)
            w = wp[:, :27].view(Cout, 3, 3, 3).permute(0, 3, 1, 2)  # [n][ky][kx][c] -> [n][c][ky][kx]
        else:
            x = inp.view(L, H, W, Cin).permute(0, 3, 1, 2)
            w = wp.view(3, 3, Cout, Cin).permute(2, 3, 0, 1)
        y = torch.relu(torch.nn.functional.conv2d(x.to(self.dtype), w.to(self.dtype), bias.to(self.dtype), padding=1))
        if pool:
            y = torch.nn.functional.max_pool2d(y, 2, 2)
        out.view(L, y.shape[2], y.shape[3], Cout).copy_(y.permute(0, 2, 3, 1).to(out.dtype))

    # ---- hl16 (fp16 hi/lo split) trunk: same conv, operands/outputs stored split-half ----------
    def conv3x3_hl16_patch(self, inp, wp, bias, out, L, H, W, Cin, Cout, pool, oscale):
        from mmmot_amd.pack import from_hl16, to_hl16
        x = from_hl16(inp.reshape(-1)[:L * H * W * Cin].view(L * H * W, Cin)).view(L, H, W, Cin).permute(0, 3, 1, 2)
        w = from_hl16(wp.reshape(9 * Cout, Cin)).view(3, 3, Cout, Cin).permute(2, 3, 0, 1)
        w = (w.double() * self._osv(oscale, Cout).reshape(Cout, 1, 1, 1))  # exact: powers of two
        y = torch.relu(torch.nn.functional.conv2d(x.to(self.dtype), w.to(self.dtype), bias.to(self.dtype), padding=1))
        if pool:
            y = torch.nn.functional.max_pool2d(y, 2, 2)
        rows = y.permute(0, 2, 3, 1).reshape(-1, Cout)
        self._guard(rows, False)
        out.reshape(-1)[:rows.numel()].view(-1, Cout).copy_(to_hl16(rows.clamp(max=65000.0)))

    def conv1_fused_hl16(self, crops, w1, bias1, oscale1, w2, bias2, oscale2, out, L, H, W):
        from mmmot_amd.pack import from_hl16, to_hl16
        w1f = from_hl16(w1.reshape(64, 32)) * oscale1
        tmp = torch.zeros(L * H * W, 64)
        self.conv3x3(crops, w1f, bias1, tmp, L, H, W, 3, 64, True, False)
        self._count(2, (tmp > 65000.0).any())
        self.conv3x3_hl16_patch(to_hl16(tmp.clamp(max=65000.0)), w2, bias2, out, L, H, W, 64, 64, True, oscale2)

    @staticmethod
    def _normalize_u8(u8, mean, std):
        """ToTensor + Normalize with the reference's fp32 operation order: [L][H][W][3] uint8 -> [L][3][H][W] fp32"""
        x = u8.permute(0, 3, 1, 2).to(torch.float32) / 255.0
        m = torch.tensor(list(mean), dtype=torch.float32).view(1, 3, 1, 1)
        sd = torch.tensor(list(std), dtype=torch.float32).view(1, 3, 1, 1)
        return ((x - m) / sd).contiguous()

    def conv1_fused_u8(self, crops_u8, mean, std, w1, bias1, oscale1, w2, bias2, oscale2, out, L, H, W, q8=False):
        x = self._normalize_u8(crops_u8.reshape(L, H, W, 3), mean, std)
        (self.conv1_fused_hq8 if q8 else self.conv1_fused_hl16)(x, w1, bias1, oscale1, w2, bias2, oscale2, out, L, H, W)

    def u8_normalize(self, crops_u8, mean_std, out, N, S):
        ms = mean_std.reshape(-1).tolist()
        out.reshape(-1)[:N * 3 * S * S].view(N, 3, S, S).copy_(self._normalize_u8(crops_u8.reshape(N, S, S, 3), ms[:3], ms[3:]))

    def _conv_hq8(self, x, wp, bias, L, H, W, Cin, Cout, pool, oscale):
        """the hq8 arithmetic: hi*hi + 2^-3 (a8 * w_lo8 + a_lo8 * w8), x = decoded parts (hi, a8, al8) NHWC"""
        from mmmot_amd.pack import hq8_parts
        wh, wl8, w8 = [t.reshape(3, 3, Cout, Cin).permute(2, 3, 0, 1).to(self.dtype) for t in hq8_parts(wp.reshape(9 * Cout, Cin))]
        ah, a8, al8 = [t.reshape(L, H, W, Cin).permute(0, 3, 1, 2).to(self.dtype) for t in x]
        cv = torch.nn.functional.conv2d
        y = cv(ah, wh, None, padding=1) + 0.125 * (cv(a8, wl8, None, padding=1) + cv(al8, w8, None, padding=1))
        y = torch.relu(y * self._osv(oscale, Cout).to(self.dtype) + bias.to(self.dtype).view(1, -1, 1, 1))
        if pool:
            y = torch.nn.functional.max_pool2d(y, 2, 2)
        return y.permute(0, 2, 3, 1).reshape(-1, Cout)

    def conv3x3_hq8(self, inp, wp, bias, out, L, H, W, Cin, Cout, pool, oscale):
        from mmmot_amd.pack import hq8_parts, to_hq8_act
        x = hq8_parts(inp.reshape(-1)[:L * H * W * Cin].view(L * H * W, Cin))
        rows = self._conv_hq8(x, wp, bias, L, H, W, Cin, Cout, pool, oscale)
        self._guard(rows, True)
        out.reshape(-1)[:rows.numel()].view(-1, Cout).copy_(to_hq8_act(rows))

    def conv1_fused_hq8(self, crops, w1, bias1, oscale1, w2, bias2, oscale2, out, L, H, W):
        from mmmot_amd.pack import from_hl16, hq8_parts, to_hq8_act
        w1f = from_hl16(w1.reshape(64, 32)) * oscale1
        tmp = torch.zeros(L * H * W, 64)
        self.conv3x3(crops, w1f, bias1, tmp, L, H, W, 3, 64, True, False)
        self._count(2, (tmp > 1792.0).any())
        rows = self._conv_hq8(hq8_parts(to_hq8_act(tmp)), w2, bias2, L, H, W, 64, 64, True, oscale2)
        self._guard(rows, True)
        out.reshape(-1)[:rows.numel()].view(-1, 64).copy_(to_hq8_act(rows))

    def hq8_pack(self, x, y):
        from mmmot_amd.pack import to_hq8_act
        y.reshape(-1).copy_(to_hq8_act(x.reshape(-1, 32)).reshape(-1))

    def hq8_unpack(self, x, y):
        from mmmot_amd.pack import from_hq8_act
        y.reshape(-1).copy_(from_hq8_act(x.reshape(-1, 32)).reshape(-1))

    def conv3x3_first_hl16(self, inp, wp, bias, out, L, H, W, Cout):
        from mmmot_amd.pack import to_hl16
        tmp = torch.zeros(L * H * W, Cout)
        self.conv3x3(inp, wp, bias, tmp, L, H, W, 3, Cout, True, False)
        out.reshape(-1)[:tmp.numel()].view(-1, Cout).copy_(to_hl16(tmp))

    def hl16_pack(self, x, y):
        from mmmot_amd.pack import to_hl16
        y.reshape(-1).copy_(to_hl16(x.reshape(-1, 8)).reshape(-1))

    def hl16_unpack(self, x, y):
        from mmmot_amd.pack import from_hl16
        y.reshape(-1).copy_(from_hl16(x.reshape(-1, 8)).reshape(-1))

    def gemm(self, W, tiles, N, K, X=None, bias=None, dbias=None, rowidx=None, Y=None, part=None,
             sc=None, sh=None, FA=None, FB=None, pair=None, amode=0, pairop=0, act=ACT_NONE,
             w_hl16=False, oscale=1.0, osc=None, osh=None, colsum=None):
        if w_hl16:  # hl16 split-half weights, pre-scaled by 1/oscale
            from mmmot_amd.pack import from_hl16
            W = from_hl16(W[:N, :K].contiguous()).double() * oscale
        R = tiles.R
        grp = _rows_groups(tiles)
        if amode == 2:
            row0, M = pair['row0'].long(), pair['M'].long()
            r = torch.arange(R)
            q = r - row0[grp]
            i, j = q // M[grp], q % M[grp]
            a = FA[pair['aoff'].long()[grp] + i, :K]
            b = FB[pair['boff'].long()[grp] + j, :K]
            A = a * b if pairop == 0 else ((a - b) / 2).abs() if pairop == 1 else (a - b) / 2
        else:
            A = X[:R, :K]
            if amode == 1:
                A = torch.relu(A * sc[grp, :K] + sh[grp, :K])
        v = A.to(self.dtype) @ W[:N, :K].to(self.dtype).t()
        if bias is not None:
            v = v + bias[:N]
        if dbias is not None:
            v = v + dbias[rowidx.long()[:R], :N]
        if part is not None:
            for t in range(tiles.T):
                r0, n = int(tiles.h_row0[t]), int(tiles.h_nrows[t])
                part[t, 0, :N] = v[r0:r0 + n].sum(0)
                part[t, 1, :N] = ((v[r0:r0 + n] - part[t, 0, :N] / n) ** 2).sum(0)
        if colsum is not None:
            w = torch.relu(v * osc[grp, :N] + osh[grp, :N])
            for t in range(tiles.T):
                r0, n = int(tiles.h_row0[t]), int(tiles.h_nrows[t])
                colsum[t, :N] = w[r0:r0 + n].sum(0).to(colsum.dtype)
        if Y is not None:
            Y[:R, :N] = _act(v, act).to(Y.dtype)

    def pn_mlp64(self, W16, oscale, tiles, N, X, sc, sh, bias, Y, part):
        self.gemm(W16, tiles, N, 64, X=X, bias=bias, Y=Y, part=part, sc=sc, sh=sh, amode=1, w_hl16=True, oscale=oscale)

    def gemm_ares(self, W16, oscale, tiles, N, K, X, sc, sh, bias=None, dbias=None, tile_dbrow=None, part=None,
                  osc=None, osh=None, colsum=None):
        from mmmot_amd.pack import from_hl16
        W = from_hl16(W16[:N, :K].contiguous()).double() * oscale
        grp = _rows_groups(tiles)
        A = torch.relu(X[:tiles.R, :K] * sc[grp, :K] + sh[grp, :K])
        v = A.to(self.dtype) @ W.to(self.dtype).t()
        if bias is not None:
            v = v + bias[:N]
        for t in range(tiles.T):
            r0, n = int(tiles.h_row0[t]), int(tiles.h_nrows[t])
            vt = v[r0:r0 + n]
            if dbias is not None:
                vt = vt + dbias[int(tile_dbrow[t]), :N]
            g = int(tiles.h_group[t])
            for h in range(2):
                vh = vt[64 * h:64 * (h + 1)]
                if part is not None:
                    part[2 * t + h, 0, :N] = vh.sum(0)
                    part[2 * t + h, 1, :N] = ((vh - vh.mean(0)) ** 2).sum(0) if vh.shape[0] else 0.0
                if colsum is not None:
                    colsum[2 * t + h, :N] = torch.relu(vh * osc[g, :N] + osh[g, :N]).sum(0).to(colsum.dtype)

    def gram_rows(self, X, K, sc, sh, tiles, Gout, Sout):
        grp = _rows_groups(tiles)
        A = torch.relu(X[:tiles.R, :K] * sc[grp, :K] + sh[grp, :K]).double()
        for t in range(tiles.T):
            r0, n = int(tiles.h_row0[t]), int(tiles.h_nrows[t])
            Gout[t] = (A[r0:r0 + n].t() @ A[r0:r0 + n]).reshape(-1)
            Sout[t] = A[r0:r0 + n].sum(0)

    def gn_finalize_gram(self, Gp, Sp, tiles, K, W, bias, N, gamma, beta, eps, work, sc, sh):
        for g in range(tiles.G):
            t0, nt, cnt = int(tiles.h_g_tile0[g]), int(tiles.h_g_ntiles[g]), float(tiles.h_g_count[g])
            m = Sp[t0:t0 + nt].double().sum(0) / cnt
            C = Gp[t0:t0 + nt].double().sum(0).view(K, K) / cnt - torch.outer(m, m)
            Wd = W[:N, :K].double()
            mean = Wd @ m + (bias[:N].double() if bias is not None else 0.0)
            var = torch.einsum('nk,kl,nl->n', Wd, C, Wd).clamp(min=0.0)
            scv = gamma[:N].double() / torch.sqrt(var + eps)
            sc[g, :N] = scv.float()
            sh[g, :N] = (beta[:N].double() - mean * scv).float()

    def gn_finalize_gram_dbias(self, Gp, Sp, tiles, tile_det, K, W, dbias, N, gamma, beta, eps, work, sc, sh):
        """executable specification of mmmot_gn_finalize_gram_dbias: statistics of v = W a + dbias[det] from the moments"""
        td = torch.as_tensor(tile_det).long()
        for g in range(tiles.G):
            t0, nt, P = int(tiles.h_g_tile0[g]), int(tiles.h_g_ntiles[g]), float(tiles.h_g_count[g])
            St = Sp[t0:t0 + nt].double()                                   # [nt][K] column sums per super-tile
            ct = torch.as_tensor(tiles.h_nrows[t0:t0 + nt]).double()       # rows per super-tile
            D = dbias[td[t0:t0 + nt], :N].double()                         # [nt][N] bias row of every super-tile
            m = St.sum(0) / P
            C = Gp[t0:t0 + nt].double().sum(0).view(K, K) / P - torch.outer(m, m)
            Wd = W[:N, :K].double()
            wm = Wd @ m
            dbar = (ct[:, None] * D).sum(0) / P
            quad = torch.einsum('nk,kl,nl->n', Wd, C, Wd)
            cross = (D * (St @ Wd.t())).sum(0) / P - wm * dbar
            vard = (ct[:, None] * D * D).sum(0) / P - dbar * dbar
            var = (quad + 2.0 * cross + vard).clamp(min=0.0)
            mean = wm + dbar
            scv = gamma[:N].double() / torch.sqrt(var + eps)
            sc[g, :N] = scv.float()
            sh[g, :N] = (beta[:N].double() - mean * scv).float()

    def gn_finalize(self, part, tiles, C, NG, gamma, beta, eps, sc, sh):
        CG = C // NG
        for g in range(tiles.G):
            t0, nt, cnt = int(tiles.h_g_tile0[g]), int(tiles.h_g_ntiles[g]), int(tiles.h_g_count[g])
            p = part[t0:t0 + nt, :, :C].double()                      # [nt][2][C]: tile sum, tile-centred M2
            n_t = torch.as_tensor(tiles.h_nrows[t0:t0 + nt]).double().view(nt, 1)
            s1 = p[:, 0].sum(0).view(NG, CG).sum(1) / (cnt * CG)      # group mean
            dev = torch.where(n_t > 0, p[:, 0] / n_t.clamp(min=1) - s1.repeat_interleave(CG), torch.zeros_like(p[:, 0]))
            m2 = (p[:, 1] + n_t * dev * dev).sum(0).view(NG, CG).sum(1)   # Chan et al. parallel combine
            var = m2 / (cnt * CG)
            rstd = 1.0 / torch.sqrt(var + eps)
            scv = gamma.double() * rstd.repeat_interleave(CG)
            sc[g, :C] = scv.float()
            sh[g, :C] = (beta.double() - s1.repeat_interleave(CG) * scv).float()

    def segment_mean(self, X, C, segs, out, sc=None, sh=None, relu=False, use_group=True, hl16=False, take_max=False):
        if int(hl16) == 2:
            from mmmot_amd.pack import from_hq8_act
            X = from_hq8_act(X[:, :C].contiguous())
        elif hl16:
            from mmmot_amd.pack import from_hl16
            X = from_hl16(X[:, :C].contiguous())
        for s in range(segs.n):
            st, cnt, stride = int(segs.h_start[s]), int(segs.h_count[s]), int(segs.h_stride[s])
            rows = X[st:st + (cnt - 1) * stride + 1:stride, :C]
            if sc is not None:
                g = int(segs.h_group[s]) if use_group else 0
                rows = rows * sc[g, :C] + sh[g, :C]
            if relu:
                rows = torch.relu(rows)
            div = getattr(segs, 'h_div', None)
            if take_max:
                out[s, :C] = rows.max(0)[0]
            else:
                out[s, :C] = rows.mean(0) if div is None else rows.sum(0) / float(div[s])

    # ---- training backward of the pairwise block (csrc/backward.hip), in executable-specification form ----
    @staticmethod
    def _row_groups(tiles):
        return _rows_groups(tiles)

    def _dz_yhat(self, dA, Y, C, sc1, sh1, gamma, beta, relu, tiles):
        grp = _rows_groups(tiles)
        R = tiles.R
        yh = Y[:R, :C].double() * sc1[grp, :C].double() + sh1[grp, :C].double()
        z = yh * gamma[:C].double() + beta[:C].double()
        dz = dA[:R, :C].double()
        if relu:
            dz = dz * (z > 0)
        return dz, yh, grp

    def gn_bwd_partial(self, dA, Y, C, sc1, sh1, gamma, beta, relu, tiles, P):
        dz, yh, _ = self._dz_yhat(dA, Y, C, sc1, sh1, gamma, beta, relu, tiles)
        for t in range(tiles.T):
            r0, n = int(tiles.h_row0[t]), int(tiles.h_nrows[t])
            P[t, 0, :C] = dz[r0:r0 + n].sum(0).float()
            P[t, 1, :C] = (dz[r0:r0 + n] * yh[r0:r0 + n]).sum(0).float()

    def gn_bwd_finalize(self, S, tiles, C, NG, gamma, M):
        S3 = S.view(tiles.G, 2, C).double()
        CG = C // NG
        for g in range(tiles.G):
            cnt = float(tiles.h_g_count[g]) * CG
            for k in range(2):
                m = (S3[g, k] * gamma[:C].double()).view(NG, CG).sum(1) / cnt
                M[g, k, :C] = m.repeat_interleave(CG).float()

    def gn_bwd_apply(self, dA, Y, C, sc1, sh1, gamma, beta, relu, M, tiles, dY):
        dz, yh, grp = self._dz_yhat(dA, Y, C, sc1, sh1, gamma, beta, relu, tiles)
        m1, m2 = M[grp, 0, :C].double(), M[grp, 1, :C].double()
        dY[:tiles.R, :C] = (sc1[grp, :C].double() * (gamma[:C].double() * dz - m1 - yh * m2)).float()

    def _a_operand(self, tiles, K, X, sc, sh, FA, FB, pair, amode, pairop):
        grp = _rows_groups(tiles)
        R = tiles.R
        if amode == 2:  # pair
            rows = []
            for g in range(len(pair['row0'])):
                N = None
            A = torch.zeros(R, K, dtype=torch.float64)
            row0, gM, aoff, boff = [t.cpu().numpy() for t in (pair['row0'], pair['M'], pair['aoff'], pair['boff'])]
            for t in range(tiles.T):
                r0, n, g = int(tiles.h_row0[t]), int(tiles.h_nrows[t]), int(tiles.h_group[t])
                local = torch.arange(r0, r0 + n) - int(row0[g])
                i, j = local // int(gM[g]), local % int(gM[g])
                a, b = FA[int(aoff[g]) + i, :K].double(), FB[int(boff[g]) + j, :K].double()
                A[r0:r0 + n] = a * b if pairop == 0 else ((a - b).abs() / 2 if pairop == 1 else (a - b) / 2)
            return A
        A = X[:R, :K].double()
        if amode == 1:
            A = torch.relu(A * sc[grp, :K].double() + sh[grp, :K].double())
        return A

    def gemm_tn(self, dY, tiles, N, K, dW, db=None, X=None, sc=None, sh=None, FA=None, FB=None, pair=None, amode=0,
                pairop=0, nsplit=1):
        A = self._a_operand(tiles, K, X, sc, sh, FA, FB, pair, amode, pairop)
        d = dY[:tiles.R, :N].double()
        dWv = dW.view(nsplit, N, K)
        dbv = None if db is None else db.view(nsplit, N)
        for s_ in range(nsplit):  # share s = tiles [T*s/nsplit, T*(s+1)/nsplit)
            t_lo, t_hi = tiles.T * s_ // nsplit, tiles.T * (s_ + 1) // nsplit
            if t_hi > t_lo:
                r_lo = int(tiles.h_row0[t_lo])
                r_hi = int(tiles.h_row0[t_hi - 1]) + int(tiles.h_nrows[t_hi - 1])
            else:
                r_lo = r_hi = 0
            dWv[s_] = (d[r_lo:r_hi].t() @ A[r_lo:r_hi]).float()
            if dbv is not None:
                dbv[s_] = d[r_lo:r_hi].sum(0).float()

    def pair_bwd(self, dX, F, dF, C, row0, gN, gM, aoff, boff, blk_group, blk_idx, pairop, side):
        for g in range(row0.numel()):
            N, M, r0, ao, bo = int(gN[g]), int(gM[g]), int(row0[g]), int(aoff[g]), int(boff[g])
            d = dX[r0:r0 + N * M, :C].double().view(N, M, C)
            a, b = F[ao:ao + N, :C].double().unsqueeze(1), F[bo:bo + M, :C].double().unsqueeze(0)
            if pairop == 0:
                wa, wb = b.expand(N, M, C), a.expand(N, M, C)
            elif pairop == 1:
                sgn = torch.sign(a - b)
                wa, wb = 0.5 * sgn, -0.5 * sgn
            else:
                wa, wb = torch.full((N, M, C), 0.5, dtype=torch.float64), torch.full((N, M, C), -0.5, dtype=torch.float64)
            if side == 0:
                dF[ao:ao + N, :C] += (d * wa).sum(1).float()
            else:
                dF[bo:bo + M, :C] += (d * wb).sum(0).float()

    def pair_expand_bwd(self, dV, dA, C, tiles, row0, gN, gM, vrow0):
        for g in range(row0.numel()):
            N, M, r0, v0 = int(gN[g]), int(gM[g]), int(row0[g]), int(vrow0[g])
            new, end = dV[v0:v0 + M, :C].double(), dV[v0 + M:v0 + M + N, :C].double()
            dA[r0:r0 + N * M, :C] = (new.unsqueeze(0) / N + end.unsqueeze(1) / M).reshape(N * M, C).float()

    def rowdot_bwd(self, X, K, w, b, sc, sh, tiles, act, gout, gidx, dA, PW):
        grp = _rows_groups(tiles)
        R = tiles.R
        a = torch.relu(X[:R, :K].double() * sc[grp, :K].double() + sh[grp, :K].double())
        pre = a @ w[:K].double() + b
        gp = (gout[gidx[:R].long()] if gidx is not None else gout[:R]).double()
        if act == ACT_SIGMOID:
            s_ = torch.sigmoid(pre)
            gp = gp * s_ * (1 - s_)
        dA[:R, :K] = (gp.unsqueeze(1) * w[:K].double().unsqueeze(0)).float()
        for t in range(tiles.T):
            r0, n = int(tiles.h_row0[t]), int(tiles.h_nrows[t])
            PW[t, :K] = (gp[r0:r0 + n].unsqueeze(1) * a[r0:r0 + n]).sum(0).float()
            PW[t, K] = gp[r0:r0 + n].sum().float()

    def softmax_pairs_bwd(self, logits, dout, dlogits, row0, gN, gM, G, max_nm, mode):
        for g in range(G):
            N, M, r0 = int(gN[g]), int(gM[g]), int(row0[g])
            with torch.enable_grad():  # called from inside an autograd.Function's backward (grad mode off there)
                x = logits[r0:r0 + N * M].double().view(N, M).clone().requires_grad_(True)
                p, q = torch.softmax(x, 1), torch.softmax(x, 0)
                out = p if mode == 1 else (p * q if mode == 2 else ((p + q) / 2 if mode == 3 else torch.max(p, q)))
            (gx,) = torch.autograd.grad(out, x, dout[r0:r0 + N * M].double().view(N, M))
            dlogits[r0:r0 + N * M] = gx.reshape(-1).float()

    def fusion_c_bwd(self, dFu, Y0, Y1, sc0, sh0, sc1, sh1, tiles, DY0, DY1, DN0, DN1, C):
        grp = _rows_groups(tiles)
        R = tiles.R
        g0, i0, g1, i1 = (Y0[:R, :C].double(), Y0[:R, C:2 * C].double(), Y1[:R, :C].double(), Y1[:R, C:2 * C].double())
        a0, a1 = torch.sigmoid(g0), torch.sigmoid(g1)
        n0 = i0 * sc0[grp, :C].double() + sh0[grp, :C].double()
        n1 = i1 * sc1[grp, :C].double() + sh1[grp, :C].double()
        den = a0 + a1
        fused = (a0 * n0 + a1 * n1) / den
        w = dFu[:R, :C].double() / den
        DN0[:R, :C], DN1[:R, :C] = (w * a0).float(), (w * a1).float()
        DY0[:R, :C] = (w * (n0 - fused) * a0 * (1 - a0)).float()
        DY1[:R, :C] = (w * (n1 - fused) * a1 * (1 - a1)).float()

    def add_rows(self, A, B, Y, C):
        Y[:, :C] = A[:, :C] + B[:, :C]

    def skippool_head(self, P, C, hd, eps, out, R):
        ln = lambda x, g, b: torch.nn.functional.layer_norm(x, (x.shape[1],), g.to(self.dtype), b.to(self.dtype), eps)
        x = ln(P[:R, :C].to(self.dtype), hd['g0'], hd['b0'])
        x = torch.relu(ln(x @ hd['w1'].to(self.dtype).t() + hd['c1'].to(self.dtype), hd['g2'], hd['b2']))
        x = torch.relu(ln(x @ hd['w4'].to(self.dtype).t() + hd['c4'].to(self.dtype), hd['g5'], hd['b5']))
        out[:R, :128] = x.to(out.dtype)

    # ---- training step, third slice (csrc/train_vgg.hip, conv3x3.hip RAW) ----
    def conv3x3_raw(self, inp, wp, bias, out, L, H, W, Cin, Cout, first):
        F = torch.nn.functional
        if first:
            x = inp.reshape(L, 3, H, W)
            w = wp[:, :27].reshape(Cout, 3, 3, 3).permute(0, 3, 1, 2)
        else:
            x = inp.reshape(-1)[:L * H * W * Cin].view(L, H, W, Cin).permute(0, 3, 1, 2)
            w = wp.view(3, 3, Cout, Cin).permute(2, 3, 0, 1)
        y = F.conv2d(x.to(self.dtype), w.to(self.dtype), bias.to(self.dtype), padding=1)
        out.reshape(-1)[:L * H * W * Cout].view(L, H, W, Cout).copy_(y.permute(0, 2, 3, 1).to(out.dtype))

    def rows_stats(self, Y, C, tiles, part):
        for t in range(tiles.T):
            r0, n = int(tiles.h_row0[t]), int(tiles.h_nrows[t])
            y = Y[r0:r0 + n, :C].to(self.dtype)
            part[t, 0, :C] = y.sum(0).to(part.dtype)
            part[t, 1, :C] = ((y - y.mean(0, keepdim=True)) ** 2).sum(0).to(part.dtype)

    def bn_relu_pool(self, Z, C, sc, sh, L, H, W, pool, A):
        a = torch.relu(Z.reshape(-1)[:L * H * W * C].view(L, H, W, C).to(self.dtype) * sc.reshape(-1)[:C].to(self.dtype) +
                       sh.reshape(-1)[:C].to(self.dtype))
        if pool:
            a = torch.nn.functional.max_pool2d(a.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
        A.reshape(-1)[:a.numel()].view(a.shape).copy_(a.to(A.dtype))

    def maxpool_bwd(self, Z, C, sc, sh, dP, L, H, W, dA):
        with torch.enable_grad():
            a = torch.relu(Z.reshape(-1)[:L * H * W * C].view(L, H, W, C).to(torch.float64) * sc.reshape(-1)[:C].double() +
                           sh.reshape(-1)[:C].double()).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
            p = torch.nn.functional.max_pool2d(a, 2, 2)
            g = dP.reshape(-1)[:p.numel()].view(L, H // 2, W // 2, C).permute(0, 3, 1, 2).double()
            (ga,) = torch.autograd.grad(p, a, g)
        dA.reshape(-1)[:L * H * W * C].view(L, H, W, C).copy_(ga.permute(0, 2, 3, 1).to(dA.dtype))

    def conv3x3_wgrad(self, dZ, A, L, H, W, Cin, Cout, nsplit, dW):
        with torch.enable_grad():
            w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
            x = A.reshape(-1)[:L * H * W * Cin].view(L, H, W, Cin).permute(0, 3, 1, 2).double()
            y = torch.nn.functional.conv2d(x, w, None, padding=1)
            g = dZ.reshape(-1)[:L * H * W * Cout].view(L, H, W, Cout).permute(0, 3, 1, 2).double()
            (gw,) = torch.autograd.grad(y, w, g)
        dW.zero_()
        dW.reshape(nsplit, 9, Cout, Cin)[0] = gw.permute(2, 3, 0, 1).reshape(9, Cout, Cin).to(dW.dtype)

    def conv3x3_first_wgrad(self, dZ, X, L, H, W, PW):
        with torch.enable_grad():
            w = torch.zeros(64, 3, 3, 3, dtype=torch.float64, requires_grad=True)
            y = torch.nn.functional.conv2d(X.reshape(L, 3, H, W).double(), w, None, padding=1)
            g = dZ.reshape(-1)[:L * H * W * 64].view(L, H, W, 64).permute(0, 3, 1, 2).double()
            (gw,) = torch.autograd.grad(y, w, g)
        PW.zero_()
        PW[0].view(64, 28)[:, :27] = gw.permute(0, 2, 3, 1).reshape(64, 27).to(PW.dtype)  # k = (ky*3+kx)*3 + colour
        PW[0].view(64, 28)[:, 27] = g.sum(dim=(0, 2, 3)).to(PW.dtype)

    # ---- training step, second slice (csrc/train.hip) ----
    def rows_gather_scale(self, S, rowidx, scale, X, C):
        idx = rowidx.long()
        v = S[idx, :C].to(self.dtype)
        if scale is not None:
            v = v * scale[idx].to(self.dtype).unsqueeze(1)
        X[:, :C] = v.to(X.dtype)

    def pointnet_layer1_bwd(self, dY, X, tiles, PW):
        K = X.shape[1]
        for t in range(tiles.T):
            r0, n = int(tiles.h_row0[t]), int(tiles.h_nrows[t])
            d, x = dY[r0:r0 + n, :64].to(self.dtype), X[r0:r0 + n].to(self.dtype)
            PW[t] = torch.cat([d.t() @ x, d.sum(0, keepdim=True).t()], dim=1).reshape(-1).to(PW.dtype)

    def score_loss(self, x, y, kind, scale, g, PL, mrow=None, mcol=None, M=0, mask_mode=0, ignore=-1.0, accumulate=False):
        R, C = x.shape
        xv, yv = x.to(self.dtype), y.to(self.dtype).reshape(1, C)
        m = torch.ones(C, dtype=self.dtype)
        if mask_mode:
            ind = (lambda v: (v == 1.0)) if mask_mode == 1 else (lambda v: (v != ignore))
            c = torch.arange(C)
            if mrow is not None:
                m = m * ind(mrow[c // M]).to(self.dtype)
            if mcol is not None:
                m = m * ind(mcol[c % M]).to(self.dtype)
        m = m.reshape(1, C)
        if kind == 0:
            l = torch.clamp(xv, min=0) - xv * yv + torch.log1p(torch.exp(-xv.abs()))
            d = torch.sigmoid(xv) - yv
        else:
            e = xv * m - yv
            if kind == 1:
                l, d = e * e, 2 * e * m
            else:
                a = e.abs()
                l = torch.where(a < 1, 0.5 * e * e, a - 0.5)
                d = torch.where(a < 1, e, torch.sign(e)) * m
        g[:, :C] = (scale * d).to(g.dtype)
        if not accumulate:
            PL.zero_()
        PL[0] += float(scale * l.sum())

    def ghm_loss(self, x, y, scale, g, PL, acc_sum, bins=30, momentum=0.75, ignore=-1.0, accumulate=False):
        """executable specification of mmmot_ghm_loss (include/mmmot_hip.h): the three passes of the kernel"""
        R, C = x.shape
        xv = x.to(torch.float32)
        yv = y.to(torch.float32).reshape(1, C).expand(R, C)
        valid = yv != ignore
        gl = (torch.sigmoid(xv) - yv).abs()
        edges = torch.tensor([i / bins for i in range(bins)] + [1.0 + 1e-6], dtype=torch.float64).to(torch.float32)
        tot = max(float(valid.sum()), 1.0)
        w = torch.zeros(R, C, dtype=torch.float32)
        used = 0
        for b in range(bins):
            inds = (gl >= edges[b]) & (gl < edges[b + 1]) & valid
            cnt = int(inds.sum())
            if cnt > 0:
                acc = float(cnt)
                if momentum > 0:
                    acc = momentum * float(acc_sum[b]) + (1.0 - momentum) * cnt
                    acc_sum[b] = acc
                w[inds] = tot / acc
                used += 1
        w = w / float(max(used, 1))
        xd, yd, wd = xv.to(self.dtype), yv.to(self.dtype), w.to(self.dtype)
        l = torch.clamp(xd, min=0) - xd * yd + torch.log1p(torch.exp(-xd.abs()))
        d = torch.sigmoid(xd) - yd
        g[:, :C] = (scale * (wd * d) / tot).to(g.dtype)
        if not accumulate:
            PL.zero_()
        PL[0] += float(scale * (wd * l).sum() / tot)

    def rowdot(self, X, K, w, b, tiles, out, sc=None, sh=None, act=ACT_NONE, use_thr=False, thr=0.0, omap=None):
        R = tiles.R
        A = X[:R, :K]
        if sc is not None:
            grp = _rows_groups(tiles)
            A = torch.relu(A * sc[grp, :K] + sh[grp, :K])
        s = _act(A @ w[:K] + b, act)
        if use_thr:
            s = s - (s < thr).float()
        if omap is not None:
            out[omap.long()[:R]] = s
        else:
            out[:R] = s

    def row_layernorm(self, X, C, gamma, beta, eps, relu, Y, R):
        x = X[:R, :C]
        mu = x.mean(1, keepdim=True)
        var = ((x - mu) ** 2).mean(1, keepdim=True)
        y = (x - mu) / torch.sqrt(var + eps) * gamma + beta
        Y[:R, :C] = torch.relu(y) if relu else y

    def pointnet_layer1(self, X, W, bias, Y, part, tiles):
        v = X @ W.t() + bias
        Y[:tiles.R] = v
        for t in range(tiles.T):
            r0, n = int(tiles.h_row0[t]), int(tiles.h_nrows[t])
            part[t, 0, :64] = v[r0:r0 + n].sum(0)
            part[t, 1, :64] = ((v[r0:r0 + n] - part[t, 0, :64] / n) ** 2).sum(0)

    def affine_act(self, X, C, sc, sh, tiles, act, Y):
        grp = _rows_groups(tiles)
        Y[:tiles.R, :C] = _act(X[:tiles.R, :C] * sc[grp, :C] + sh[grp, :C], act)

    def fusion_combine(self, mode, cat, Y0, Y1, sc0, sh0, sc1, sh1, tiles, F, Lt, C):
        grp = _rows_groups(tiles)
        F[0] = cat[:, :C]
        F[1] = cat[:, C:2 * C]
        n0 = lambda: Y0[:, -C:] * sc0[grp, :C] + sh0[grp, :C]
        n1 = lambda: Y1[:, -C:] * sc1[grp, :C] + sh1[grp, :C]
        if mode == 0:
            F[2] = Y0[:, :C] * sc0[grp, :C] + sh0[grp, :C]
        elif mode == 1:
            F[2] = (Y0[:, :C] * sc0[grp, :C] + sh0[grp, :C]) + (Y1[:, :C] * sc1[grp, :C] + sh1[grp, :C])
        else:
            g0, g1 = torch.sigmoid(Y0[:, :C]), torch.sigmoid(Y1[:, :C])
            F[2] = (g0 * n0() + g1 * n1()) / (g0 + g1)

    def softmax_pairs(self, logits, out, row0, gN, gM, G, max_nm, mode):
        for g in range(G):
            r0, N, M = int(row0[g]), int(gN[g]), int(gM[g])
            assert N + M <= max_nm
            x = logits[r0:r0 + N * M].view(N, M)
            p = torch.softmax(x, dim=1)
            if mode == 1:
                r = p
            else:
                q = torch.softmax(x, dim=0)
                r = p * q if mode == 2 else (p + q) / 2 if mode == 3 else torch.max(p, q)
            out[r0:r0 + N * M] = r.reshape(-1)

    def selftest_mfma(self, A, B, C, K):
        C.copy_(A.view(32, K) @ B.view(32, K).t())

"""Python-side operator layer over the C-ABI (include/mmmot_hip.h).

``HipOps`` takes torch CUDA tensors (possibly 2-D row-strided views), extracts
raw device pointers / leading dimensions and calls libmmmot_hip.so on torch's
current HIP stream.  torch is used for device memory and streams only - no
torch operator computes anything on this path.  CPU tensors are rejected:
there is no fallback.
"""
import contextlib
import ctypes
import os
import threading

import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2
A_PLAIN, A_NORM_RELU, A_PAIR = 0, 1, 2
PAIR_OPS = {'multiply': 0, 'minus_abs': 1, 'minus': 2}
SOFTMAX_MODES = {'single': 1, 'dual': 2, 'dual_add': 3, 'dual_max': 4}
FUSION_MODES = {'A': 0, 'B': 1, 'C': 2}
LOSS_KINDS = {'bce': 0, 'l2': 1, 'l1': 2}


def _ptr(t, dtype=torch.float32):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('mmmot_amd HIP ops need device tensors (got %s); there is no CPU fallback' % t.device)
    if t.dtype != dtype:
        raise TypeError('expected %s, got %s' % (dtype, t.dtype))
    return t.data_ptr()


def _iptr(t):
    return _ptr(t, torch.int32)


def _ld(t):
    """Leading dimension of a 2-D row-strided view (unit inner stride)."""
    if t is None:
        return 0
    if t.dim() == 1:
        return t.shape[0]
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError('expected a 2-D view with unit inner stride, got shape %s stride %s' % (
            tuple(t.shape), tuple(t.stride())))
    return t.stride(0)


class HipOps:
    """The product backend: every method is one C-ABI call."""
    name = 'hip'

    def __init__(self):
        self.lib = _lib.load()
        self._osvecs = {}
        self._scr = {}
        # weight-gradient GEMMs: fp16 matrix cores with the 3-term split (default) or MMMOT_GEMM_TN=f32: exact fp32 MFMA
        self.tn_f16 = os.environ.get('MMMOT_GEMM_TN', 'f16x3') != 'f32'
        # the stream a launch sequence was pinned to is per THREAD: two threads sharing one model, each under its own
        # torch.cuda.stream(), must not see each other's pin
        self._tls = threading.local()

    @property
    def _pinned_stream(self):
        return getattr(self._tls, 'stream', None)

    @_pinned_stream.setter
    def _pinned_stream(self, v):
        self._tls.stream = v

    def _scratch(self, name, n, device):
        key = (name, n, str(device))
        t = self._scr.get(key)
        if t is None:
            t = self._scr[key] = torch.empty(n, dtype=torch.float32, device=device)
        return t

    def _osv(self, oscale, Cout, like):
        """The [Cout] per-output-channel scale vector of the trunk entry points; a python float (the kernel tests'
        one-scale-per-layer case) is expanded to a cached constant vector."""
        if torch.is_tensor(oscale):
            if oscale.numel() != Cout:
                raise ValueError('oscale must hold one scale per output channel')
            return oscale
        key = (float(oscale), Cout, like.device)
        v = self._osvecs.get(key)
        if v is None:
            v = self._osvecs[key] = torch.full((Cout,), float(oscale), dtype=torch.float32, device=like.device)
        return v

    def _stream(self):
        if self._pinned_stream is not None:
            return self._pinned_stream
        return torch.cuda.current_stream().cuda_stream

    @contextlib.contextmanager
    def on_current_stream(self, stream=None):
        """Resolve torch's current HIP stream once for a whole launch sequence (``torch.cuda.current_stream()`` costs
        ~8 us per call - 0.6 ms over the ~80 launches of a forward); the stream must not change inside the block.
        ``stream``: the already resolved current stream (a torch.cuda.Stream), if the caller has it."""
        prev = self._pinned_stream
        self._pinned_stream = (stream if stream is not None else torch.cuda.current_stream()).cuda_stream
        try:
            yield
        finally:
            self._pinned_stream = prev

    @contextlib.contextmanager
    def on_stream(self, stream):
        """Launch on ``stream`` (a torch.cuda.Stream) inside the block - the LiDAR branch of a two-stream forward -
        whatever stream the surrounding ``on_current_stream`` block pinned."""
        prev = self._pinned_stream
        self._pinned_stream = stream.cuda_stream
        try:
            with torch.cuda.stream(stream):
                yield
        finally:
            self._pinned_stream = prev

    def conv3x3(self, inp, wp, bias, out, L, H, W, Cin, Cout, first, pool):
        st = self.lib.mmmot_conv3x3_bn_relu(_ptr(inp), _ptr(wp), _ptr(bias), _ptr(out), L, H, W, Cin, Cout,
                                            int(first), int(pool), self._stream())
        _lib.check(st, 'mmmot_conv3x3_bn_relu')

    def conv3x3_hl16_patch(self, inp, wp, bias, out, L, H, W, Cin, Cout, pool, oscale):
        """fp16-split trunk layer (LDS-resident haloed patch kernel): inp/out/wp are fp32-typed buffers holding hl16
        data (same bytes), oscale the [Cout] vector of per-output-channel 2^-shift."""
        st = self.lib.mmmot_conv3x3_bn_relu_hl16_patch(_ptr(inp), _ptr(wp), _ptr(bias), _ptr(out), L, H, W, Cin,
                                                       Cout, int(pool), _ptr(self._osv(oscale, Cout, out)), self._stream())
        _lib.check(st, 'mmmot_conv3x3_bn_relu_hl16_patch')

    def conv1_fused_hl16(self, crops, w1, bias1, oscale1, w2, bias2, oscale2, out, L, H, W):
        """conv1_1 + conv1_2 + max-pool in one launch (the conv1_1 tensor never reaches HBM)."""
        st = self.lib.mmmot_conv1_fused_hl16(_ptr(crops), _ptr(w1), _ptr(bias1), float(oscale1), _ptr(w2),
                                             _ptr(bias2), _ptr(self._osv(oscale2, 64, out)), _ptr(out), L, H, W, self._stream())
        _lib.check(st, 'mmmot_conv1_fused_hl16')

    def conv3x3_hq8(self, inp, wp, bias, out, L, H, W, Cin, Cout, pool, oscale):
        """conv3x3_hl16_patch in hq8 arithmetic: activations / weights are hq8 records (pack.to_hq8_act / to_hq8_w)."""
        st = self.lib.mmmot_conv3x3_bn_relu_hq8(_ptr(inp), _ptr(wp), _ptr(bias), _ptr(out), L, H, W, Cin, Cout,
                                                int(pool), _ptr(self._osv(oscale, Cout, out)), self._stream())
        _lib.check(st, 'mmmot_conv3x3_bn_relu_hq8')

    def conv1_fused_hq8(self, crops, w1, bias1, oscale1, w2, bias2, oscale2, out, L, H, W):
        """conv1_fused_hl16 with conv1_2 in hq8 arithmetic (w1 hl16, w2 hq8, out hq8)."""
        st = self.lib.mmmot_conv1_fused_hq8(_ptr(crops), _ptr(w1), _ptr(bias1), float(oscale1), _ptr(w2),
                                            _ptr(bias2), _ptr(self._osv(oscale2, 64, out)), _ptr(out), L, H, W, self._stream())
        _lib.check(st, 'mmmot_conv1_fused_hq8')

    def conv1_fused_u8(self, crops_u8, mean, std, w1, bias1, oscale1, w2, bias2, oscale2, out, L, H, W, q8=False):
        """conv1_fused_hl16 / _hq8 fed by the 8-bit crops [L][H][W][3]: ToTensor + Normalize in the loader"""
        st = self.lib.mmmot_conv1_fused_u8(_ptr(crops_u8, torch.uint8), float(mean[0]), float(mean[1]), float(mean[2]),
                                           float(std[0]), float(std[1]), float(std[2]), _ptr(w1), _ptr(bias1),
                                           float(oscale1), _ptr(w2), _ptr(bias2), _ptr(self._osv(oscale2, 64, out)),
                                           _ptr(out), L, H, W, int(bool(q8)), self._stream())
        _lib.check(st, 'mmmot_conv1_fused_u8')

    def u8_normalize(self, crops_u8, mean_std, out, N, S):
        st = self.lib.mmmot_u8_normalize(_ptr(crops_u8, torch.uint8), N, S, _ptr(mean_std), _ptr(out), self._stream())
        _lib.check(st, 'mmmot_u8_normalize')

    def trunk_range_read(self, device, reset=True):
        """(e4m3-saturated, fp16-clamped, conv1_1 hits, 0) activation-element counters of the trunk epilogues on
        ``device`` since the last reset; synchronises the current stream (see mmmot_trunk_range_read)."""
        if torch.device(device).type != 'cuda':
            raise RuntimeError('mmmot_amd HIP ops need device tensors (got %s); there is no CPU fallback' % device)
        buf = (ctypes.c_uint * 4)()
        with torch.cuda.device(device):
            torch.cuda.current_stream().synchronize()
            _lib.check(self.lib.mmmot_trunk_range_read(buf, int(reset)), 'mmmot_trunk_range_read')
        return tuple(int(v) for v in buf)

    def trunk_range_bind(self, counters):
        """Counter block (int32 [4] device tensor, or None = the library's per-device block) the trunk launches issued
        by this thread report to from now on (mmmot_trunk_range_bind)."""
        _lib.check(self.lib.mmmot_trunk_range_bind(None if counters is None else _ptr(counters, torch.int32)),
                   'mmmot_trunk_range_bind')

    def hq8_pack(self, x, y):
        _lib.check(self.lib.mmmot_hq8_pack(_ptr(x), _ptr(y), x.numel(), self._stream()), 'mmmot_hq8_pack')

    def hq8_unpack(self, x, y):
        _lib.check(self.lib.mmmot_hq8_unpack(_ptr(x), _ptr(y), y.numel(), self._stream()), 'mmmot_hq8_unpack')

    def conv3x3_first_hl16(self, inp, wp, bias, out, L, H, W, Cout):
        st = self.lib.mmmot_conv3x3_first_hl16(_ptr(inp), _ptr(wp), _ptr(bias), _ptr(out), L, H, W, Cout,
                                               self._stream())
        _lib.check(st, 'mmmot_conv3x3_first_hl16')

    def hl16_pack(self, x, y):
        _lib.check(self.lib.mmmot_hl16_pack(_ptr(x), _ptr(y), x.numel(), self._stream()), 'mmmot_hl16_pack')

    def hl16_unpack(self, x, y):
        _lib.check(self.lib.mmmot_hl16_unpack(_ptr(x), _ptr(y), y.numel(), self._stream()), 'mmmot_hl16_unpack')

    # ---- training step on the fp16 matrix cores: device-side power-of-two scales, raw-output trunk convolution ----
    def absmax(self, X, out):
        """out[0] = max |X| over a contiguous [R][C] tensor (device scalar, no host round trip)"""
        R, C = X.numel() // X.shape[-1], X.shape[-1]
        _lib.check(self.lib.mmmot_absmax(_ptr(X), C, R, C, _ptr(out), self._stream()), 'mmmot_absmax')

    def hl16_pack_pow2(self, x, y, amax, target):
        _lib.check(self.lib.mmmot_hl16_pack_pow2(_ptr(x), _ptr(y), x.numel(), _ptr(amax), int(target), self._stream()),
                   'mmmot_hl16_pack_pow2')

    def pow2_oscale(self, out, amax_a, target_a, amax_b=None, target_b=0):
        _lib.check(self.lib.mmmot_pow2_oscale(_ptr(out), out.numel(), _ptr(amax_a), int(target_a), _ptr(amax_b), int(target_b),
                                              self._stream()), 'mmmot_pow2_oscale')

    def conv3x3_raw_hl16(self, x16, w16, bias, out, L, H, W, Cin, Cout, oscale):
        st = self.lib.mmmot_conv3x3_raw_hl16(_ptr(x16), _ptr(w16), _ptr(bias), _ptr(out), L, H, W, Cin, Cout, _ptr(oscale),
                                             self._stream())
        _lib.check(st, 'mmmot_conv3x3_raw_hl16')

    def gemm(self, W, tiles, N, K, X=None, bias=None, dbias=None, rowidx=None, Y=None, part=None,
             sc=None, sh=None, FA=None, FB=None, pair=None, amode=A_PLAIN, pairop=0, act=ACT_NONE,
             w_hl16=False, oscale=1.0, osc=None, osh=None, colsum=None):
        """W: [N][K] fp32 weights, or (w_hl16=True) the hl16 split-half copy scaled by 1/oscale.
        colsum [T][N] (+ osc/osh [G][N]): per-tile column sums of relu(v * osc + osh) (fused next-norm consumer)."""
        a = _lib.GemmArgs()
        a.X, a.ldx = _ptr(X), _ld(X)
        a.W = _ptr(W)
        a.bias = _ptr(bias)
        a.dbias, a.rowidx, a.lddb = _ptr(dbias), _iptr(rowidx), _ld(dbias)
        a.Y, a.ldy = _ptr(Y), _ld(Y)
        a.part = _ptr(part)
        a.sc, a.sh, a.ldsc = _ptr(sc), _ptr(sh), _ld(sc)
        a.FA, a.FB, a.ldf = _ptr(FA), _ptr(FB), _ld(FA)
        a.tile_row0, a.tile_nrows, a.tile_group = _iptr(tiles.row0), _iptr(tiles.nrows), _iptr(tiles.group)
        if pair is not None:
            a.grp_row0, a.grp_M = _iptr(pair['row0']), _iptr(pair['M'])
            a.grp_aoff, a.grp_boff = _iptr(pair['aoff']), _iptr(pair['boff'])
            a.pair_uniform32 = int(bool(pair.get('uniform32', False)))  # every M a multiple of 32 (the caller's plan knows)
        a.T, a.N, a.K = tiles.T, N, K
        a.amode, a.pairop, a.act = amode, pairop, act
        a.w_hl16, a.oscale = int(w_hl16), float(oscale)
        a.osc, a.osh, a.ldosc = _ptr(osc), _ptr(osh), _ld(osc)
        a.colsum = _ptr(colsum)
        _lib.check(self.lib.mmmot_gemm_rows(ctypes.byref(a), self._stream()), 'mmmot_gemm_rows')

    def gemm_ares(self, W16, oscale, tiles, N, K, X, sc, sh, bias=None, dbias=None, tile_dbrow=None, part=None,
                  osc=None, osh=None, colsum=None):
        """A-resident GEMM (hl16 weights): statistics `part` [2T][2][N] and/or `colsum` [2T][N] per 64-row
        half tile; the product itself is never stored (see include/mmmot_hip.h: mmmot_gemm_ares)."""
        a = _lib.GemmAresArgs()
        a.X, a.ldx = _ptr(X), _ld(X)
        a.sc, a.sh, a.ldsc = _ptr(sc), _ptr(sh), _ld(sc)
        a.W, a.bias = _ptr(W16), _ptr(bias)
        a.dbias, a.tile_dbrow, a.lddb = _ptr(dbias), _iptr(tile_dbrow), _ld(dbias)
        a.tile_row0, a.tile_nrows, a.tile_group = _iptr(tiles.row0), _iptr(tiles.nrows), _iptr(tiles.group)
        a.part = _ptr(part)
        a.osc, a.osh, a.ldosc = _ptr(osc), _ptr(osh), _ld(osc)
        a.colsum = _ptr(colsum)
        a.T, a.N, a.K, a.oscale = tiles.T, N, K, float(oscale)
        _lib.check(self.lib.mmmot_gemm_ares(ctypes.byref(a), self._stream()), 'mmmot_gemm_ares')

    def pn_mlp64(self, W16, oscale, tiles, N, X, sc, sh, bias, Y, part):
        """PointNet K = 64 layer: Y / part of relu(X*sc+sh) W^T + b from the persistent weight-resident kernel (same
        results contract as gemm(amode=A_NORM_RELU, w_hl16=True, K=64); see mmmot_pn_mlp64)."""
        st = self.lib.mmmot_pn_mlp64(_ptr(X), _ld(X), _ptr(sc), _ptr(sh), _ld(sc), _ptr(W16), float(oscale),
                                     _ptr(bias), _ptr(Y), _ld(Y), _ptr(part), _iptr(tiles.row0), _iptr(tiles.nrows),
                                     _iptr(tiles.group), tiles.T, N, self._stream())
        _lib.check(st, 'mmmot_pn_mlp64')

    def gram_rows(self, X, K, sc, sh, tiles, Gout, Sout):
        """Per super-tile Gram matrix / column sums (float64) of relu(X*sc+sh); see mmmot_gram_rows."""
        st = self.lib.mmmot_gram_rows(_ptr(X), _ld(X), K, _ptr(sc), _ptr(sh), _ld(sc), _iptr(tiles.row0),
                                      _iptr(tiles.nrows), _iptr(tiles.group), tiles.T,
                                      _ptr(Gout, torch.float64), _ptr(Sout, torch.float64), self._stream())
        _lib.check(st, 'mmmot_gram_rows')

    def gn_finalize_gram(self, Gp, Sp, tiles, K, W, bias, N, gamma, beta, eps, work, sc, sh):
        """GroupNorm(N, N) scale/shift of v = W a + b from the Gram partials of a; see mmmot_gn_finalize_gram."""
        st = self.lib.mmmot_gn_finalize_gram(_ptr(Gp, torch.float64), _ptr(Sp, torch.float64), _iptr(tiles.g_tile0),
                                             _iptr(tiles.g_ntiles), _iptr(tiles.g_count), tiles.G, K, _ptr(W),
                                             _ptr(bias), N, _ptr(gamma), _ptr(beta), float(eps),
                                             _ptr(work, torch.float64), _ptr(sc), _ptr(sh), self._stream())
        _lib.check(st, 'mmmot_gn_finalize_gram')

    def gn_finalize_gram_dbias(self, Gp, Sp, tiles, tile_det, K, W, dbias, N, gamma, beta, eps, work, sc, sh):
        """GroupNorm(N, N) scale/shift of v = W a + dbias[det] from the Gram partials of a over detection-aligned
        super-tiles (tile_det: int32 [T], the dbias row of every super-tile); see mmmot_gn_finalize_gram_dbias."""
        st = self.lib.mmmot_gn_finalize_gram_dbias(_ptr(Gp, torch.float64), _ptr(Sp, torch.float64), _iptr(tiles.g_tile0),
                                                   _iptr(tiles.g_ntiles), _iptr(tiles.g_count), tiles.G, _iptr(tiles.nrows),
                                                   _iptr(tile_det), K, _ptr(W), _ptr(dbias), _ld(dbias), N, _ptr(gamma),
                                                   _ptr(beta), float(eps), _ptr(work, torch.float64), _ptr(sc), _ptr(sh),
                                                   self._stream())
        _lib.check(st, 'mmmot_gn_finalize_gram_dbias')

    def gn_finalize(self, part, tiles, C, NG, gamma, beta, eps, sc, sh):
        """part: [T][2][>=C] view (unit inner stride); statistics of its first C channels."""
        if part.dim() != 3 or part.stride(2) != 1 or part.stride(0) != 2 * part.stride(1):
            raise ValueError('part must be a [T][2][C] view of a [T][2][ldp] buffer')
        st = self.lib.mmmot_gn_finalize(_ptr(part), _iptr(tiles.g_tile0), _iptr(tiles.g_ntiles),
                                        _iptr(tiles.g_count), _iptr(tiles.nrows) if tiles.ragged else None, tiles.G,
                                        part.stride(1), C, NG, _ptr(gamma), _ptr(beta),
                                        float(eps), _ptr(sc), _ptr(sh), self._stream())
        _lib.check(st, 'mmmot_gn_finalize')

    def segment_mean(self, X, C, segs, out, sc=None, sh=None, relu=False, use_group=True, hl16=False, take_max=False):
        """segs.div (optional int32 tensor): divisor per segment instead of its row count; take_max: maximum over the
        segment instead of the mean."""
        st = self.lib.mmmot_segment_mean(_ptr(X), _ld(X), C, _iptr(segs.start), _iptr(segs.count),
                                         _iptr(segs.stride), _iptr(segs.group) if use_group else None,
                                         _iptr(getattr(segs, 'div', None)), segs.n,
                                         _ptr(sc), _ptr(sh), _ld(sc), int(bool(relu)) | (2 if take_max else 0), _ptr(out), _ld(out),
                                         int(hl16), self._stream())
        _lib.check(st, 'mmmot_segment_mean')

    def rowdot(self, X, K, w, b, tiles, out, sc=None, sh=None, act=ACT_NONE, use_thr=False, thr=0.0, omap=None):
        st = self.lib.mmmot_rowdot(_ptr(X), _ld(X), K, _ptr(w), float(b), _ptr(sc), _ptr(sh), _ld(sc),
                                   _iptr(tiles.row0), _iptr(tiles.nrows), _iptr(tiles.group), tiles.T, act,
                                   int(use_thr), float(thr), _ptr(out), _iptr(omap), self._stream())
        _lib.check(st, 'mmmot_rowdot')

    def row_layernorm(self, X, C, gamma, beta, eps, relu, Y, R):
        st = self.lib.mmmot_row_layernorm(_ptr(X), _ld(X), C, _ptr(gamma), _ptr(beta), float(eps), int(relu),
                                          _ptr(Y), _ld(Y), R, self._stream())
        _lib.check(st, 'mmmot_row_layernorm')

    def skippool_head(self, P, C, hd, eps, out, R):
        """one SkipPool head (appear_net.py:19-32) on the pooled rows P [R][C] -> out [R][128]; hd: packed head dict"""
        C4 = int(hd['w1'].shape[0])
        st = self.lib.mmmot_skippool_head(_ptr(P), _ld(P), C, C4, _ptr(hd['g0']), _ptr(hd['b0']), _ptr(hd['w1']),
                                          _ptr(hd['c1']), _ptr(hd['g2']), _ptr(hd['b2']), _ptr(hd['w4']), _ptr(hd['c4']),
                                          _ptr(hd['g5']), _ptr(hd['b5']), float(eps), _ptr(out), _ld(out), R, self._stream())
        _lib.check(st, 'mmmot_skippool_head')

    def pointnet_layer1(self, X, W, bias, Y, part, tiles):
        """X [P][K] with K = W.shape[1] in (3, 4)."""
        st = self.lib.mmmot_pointnet_layer1(_ptr(X), int(W.shape[1]), _ptr(W), _ptr(bias), _ptr(Y), _ptr(part),
                                            _iptr(tiles.row0), _iptr(tiles.nrows), tiles.T, self._stream())
        _lib.check(st, 'mmmot_pointnet_layer1')

    def affine_act(self, X, C, sc, sh, tiles, act, Y):
        st = self.lib.mmmot_affine_act(_ptr(X), _ld(X), C, _ptr(sc), _ptr(sh), _ld(sc), _iptr(tiles.row0),
                                       _iptr(tiles.nrows), _iptr(tiles.group), tiles.T, act, _ptr(Y), _ld(Y),
                                       self._stream())
        _lib.check(st, 'mmmot_affine_act')

    def fusion_combine(self, mode, cat, Y0, Y1, sc0, sh0, sc1, sh1, tiles, F, Lt, C):
        st = self.lib.mmmot_fusion_combine(mode, _ptr(cat), _ptr(Y0), _ld(Y0), _ptr(Y1), _ld(Y1), _ptr(sc0),
                                           _ptr(sh0), _ptr(sc1), _ptr(sh1), _ld(sc0), _iptr(tiles.row0),
                                           _iptr(tiles.nrows), _iptr(tiles.group), tiles.T, _ptr(F), Lt, C,
                                           self._stream())
        _lib.check(st, 'mmmot_fusion_combine')

    def softmax_pairs(self, logits, out, row0, gN, gM, G, max_nm, mode):
        st = self.lib.mmmot_softmax_pairs(_ptr(logits), _ptr(out), _iptr(row0), _iptr(gN), _iptr(gM), G,
                                          max_nm, mode, self._stream())
        _lib.check(st, 'mmmot_softmax_pairs')

    # ---- training backward of the pairwise block (include/mmmot_hip.h, csrc/backward.hip) -------------------
    def gn_bwd_partial(self, dA, Y, C, sc1, sh1, gamma, beta, relu, tiles, P):
        st = self.lib.mmmot_gn_bwd_partial(_ptr(dA), _ld(dA), _ptr(Y), _ld(Y), C, _ptr(sc1), _ptr(sh1), _ld(sc1),
                                           _ptr(gamma), _ptr(beta), int(relu), _iptr(tiles.row0), _iptr(tiles.nrows),
                                           _iptr(tiles.group), tiles.T, _ptr(P), self._stream())
        _lib.check(st, 'mmmot_gn_bwd_partial')

    def gn_bwd_finalize(self, S, tiles, C, NG, gamma, M):
        st = self.lib.mmmot_gn_bwd_finalize(_ptr(S), _iptr(tiles.g_count), tiles.G, C, NG, _ptr(gamma), _ptr(M),
                                            self._stream())
        _lib.check(st, 'mmmot_gn_bwd_finalize')

    def gn_bwd_apply(self, dA, Y, C, sc1, sh1, gamma, beta, relu, M, tiles, dY):
        st = self.lib.mmmot_gn_bwd_apply(_ptr(dA), _ld(dA), _ptr(Y), _ld(Y), C, _ptr(sc1), _ptr(sh1), _ld(sc1),
                                         _ptr(gamma), _ptr(beta), int(relu), _ptr(M), _iptr(tiles.row0),
                                         _iptr(tiles.nrows), _iptr(tiles.group), tiles.T, _ptr(dY), _ld(dY),
                                         self._stream())
        _lib.check(st, 'mmmot_gn_bwd_apply')

    def gemm_tn(self, dY, tiles, N, K, dW, db=None, X=None, sc=None, sh=None, FA=None, FB=None, pair=None,
                amode=A_PLAIN, pairop=0, nsplit=1):
        """dW [nsplit][N][K] / db [nsplit][N]: per-share partial sums when nsplit > 1 (see mmmot_gemm_tn)."""
        a = _lib.GemmTnArgs()
        a.nsplit = int(nsplit)
        a.dY, a.lddy = _ptr(dY), _ld(dY)
        a.X, a.ldx = _ptr(X), _ld(X)
        a.sc, a.sh, a.ldsc = _ptr(sc), _ptr(sh), _ld(sc)
        a.FA, a.FB, a.ldf = _ptr(FA), _ptr(FB), _ld(FA)
        a.tile_row0, a.tile_nrows, a.tile_group = _iptr(tiles.row0), _iptr(tiles.nrows), _iptr(tiles.group)
        if pair is not None:
            a.grp_row0, a.grp_M = _iptr(pair['row0']), _iptr(pair['M'])
            a.grp_aoff, a.grp_boff = _iptr(pair['aoff']), _iptr(pair['boff'])
        a.T, a.N, a.K, a.amode, a.pairop = tiles.T, N, K, amode, pairop
        a.dW, a.db = _ptr(dW), _ptr(db)
        if self.tn_f16:
            # fp16 matrix cores (3-term split), dY scaled by a power of two taken from its maximum on the device
            # one scalar per (device, stream): two backward passes issued on different streams through one HipOps
            # must not land their maxima in each other's scale
            amax = self._scratch(('tn_amax', int(self._stream() or 0)), 1, dY.device)
            _lib.check(self.lib.mmmot_absmax(_ptr(dY), _ld(dY), tiles.R, N, _ptr(amax), self._stream()), 'mmmot_absmax')
            _lib.check(self.lib.mmmot_gemm_tn_f16(ctypes.byref(a), _ptr(amax), self._stream()), 'mmmot_gemm_tn_f16')
            return
        _lib.check(self.lib.mmmot_gemm_tn(ctypes.byref(a), self._stream()), 'mmmot_gemm_tn')

    def pair_bwd(self, dX, F, dF, C, row0, gN, gM, aoff, boff, blk_group, blk_idx, pairop, side):
        st = self.lib.mmmot_pair_bwd(_ptr(dX), _ld(dX), _ptr(F), _ld(F), _ptr(dF), _ld(dF), C, _iptr(row0), _iptr(gN),
                                     _iptr(gM), _iptr(aoff), _iptr(boff), _iptr(blk_group), _iptr(blk_idx),
                                     blk_group.numel(), pairop, side, self._stream())
        _lib.check(st, 'mmmot_pair_bwd')

    def pair_expand_bwd(self, dV, dA, C, tiles, row0, gN, gM, vrow0):
        st = self.lib.mmmot_pair_expand_bwd(_ptr(dV), _ld(dV), _ptr(dA), _ld(dA), C, _iptr(tiles.row0),
                                            _iptr(tiles.nrows), _iptr(tiles.group), tiles.T, _iptr(row0), _iptr(gN),
                                            _iptr(gM), _iptr(vrow0), self._stream())
        _lib.check(st, 'mmmot_pair_expand_bwd')

    def rowdot_bwd(self, X, K, w, b, sc, sh, tiles, act, gout, gidx, dA, PW):
        st = self.lib.mmmot_rowdot_bwd(_ptr(X), _ld(X), K, _ptr(w), float(b), _ptr(sc), _ptr(sh), _ld(sc),
                                       _iptr(tiles.row0), _iptr(tiles.nrows), _iptr(tiles.group), tiles.T, act,
                                       _ptr(gout), _iptr(gidx), _ptr(dA), _ld(dA), _ptr(PW), _ld(PW), self._stream())
        _lib.check(st, 'mmmot_rowdot_bwd')

    def softmax_pairs_bwd(self, logits, dout, dlogits, row0, gN, gM, G, max_nm, mode):
        st = self.lib.mmmot_softmax_pairs_bwd(_ptr(logits), _ptr(dout), _ptr(dlogits), _iptr(row0), _iptr(gN),
                                              _iptr(gM), G, max_nm, mode, self._stream())
        _lib.check(st, 'mmmot_softmax_pairs_bwd')

    def fusion_c_bwd(self, dFu, Y0, Y1, sc0, sh0, sc1, sh1, tiles, DY0, DY1, DN0, DN1, C):
        st = self.lib.mmmot_fusion_c_bwd(_ptr(dFu), _ptr(Y0), _ld(Y0), _ptr(Y1), _ld(Y1), _ptr(sc0), _ptr(sh0),
                                         _ptr(sc1), _ptr(sh1), _ld(sc0), _iptr(tiles.row0), _iptr(tiles.nrows),
                                         _iptr(tiles.group), tiles.T, _ptr(DY0), _ptr(DY1), _ld(DY0), _ptr(DN0),
                                         _ptr(DN1), C, self._stream())
        _lib.check(st, 'mmmot_fusion_c_bwd')

    def add_rows(self, A, B, Y, C):
        st = self.lib.mmmot_add_rows(_ptr(A), _ld(A), _ptr(B), _ld(B), _ptr(Y), _ld(Y), A.shape[0], C, self._stream())
        _lib.check(st, 'mmmot_add_rows')

    # ---- training step, second slice (csrc/train.hip) ----------------------------------------------------------
    def rows_gather_scale(self, S, rowidx, scale, X, C):
        """X[r][:C] = S[rowidx[r]][:C] * scale[rowidx[r]] (scale None: 1): backward of the per-detection average pools"""
        st = self.lib.mmmot_rows_gather_scale(_ptr(S), _ld(S), _iptr(rowidx), _ptr(scale), _ptr(X), _ld(X), X.shape[0], C,
                                              self._stream())
        _lib.check(st, 'mmmot_rows_gather_scale')

    def pointnet_layer1_bwd(self, dY, X, tiles, PW):
        """PW [T][64 * (K + 1)] per-tile partials of (dW1 [64][K] | db1 [64]) interleaved per channel; K = X.shape[1]"""
        st = self.lib.mmmot_pointnet_layer1_bwd(_ptr(dY), _ptr(X), int(X.shape[1]), _iptr(tiles.row0), _iptr(tiles.nrows),
                                                tiles.T, _ptr(PW), self._stream())
        _lib.check(st, 'mmmot_pointnet_layer1_bwd')

    def score_loss(self, x, y, kind, scale, g, PL, mrow=None, mcol=None, M=0, mask_mode=0, ignore=-1.0, accumulate=False):
        """one TrackingLoss term over x [R][C] (see mmmot_score_loss); PL [nblocks] partial sums of the scaled loss
        (accumulate: added to PL's contents)"""
        R, C = x.shape
        st = self.lib.mmmot_score_loss(_ptr(x), _ld(x), _ptr(y), _ptr(mrow), _ptr(mcol), int(M), int(mask_mode),
                                       float(ignore), int(kind), float(scale), R, C, _ptr(g), _ld(g), _ptr(PL),
                                       PL.numel(), int(bool(accumulate)), self._stream())
        _lib.check(st, 'mmmot_score_loss')

    # ---- training step, third slice: training-mode VGG trunk (csrc/train_vgg.hip, conv3x3.hip RAW) ----
    def ghm_loss(self, x, y, scale, g, PL, acc_sum, bins=30, momentum=0.75, ignore=-1.0, accumulate=False):
        """the 'ghm' DetLoss term (mmmot_ghm_loss): x, g [R][C]; y [C]; acc_sum float64 [bins] = the module's running state"""
        R, C = x.shape
        st = self.lib.mmmot_ghm_loss(_ptr(x), _ld(x), _ptr(y), float(ignore), float(scale), R, C, int(bins), float(momentum),
                                     _ptr(acc_sum, torch.float64), _ptr(g), _ld(g), _ptr(PL), int(accumulate), self._stream())
        _lib.check(st, 'mmmot_ghm_loss')

    def conv3x3_raw(self, inp, wp, bias, out, L, H, W, Cin, Cout, first):
        st = self.lib.mmmot_conv3x3_raw(_ptr(inp), _ptr(wp), _ptr(bias), _ptr(out), L, H, W, Cin, Cout, int(first),
                                        self._stream())
        _lib.check(st, 'mmmot_conv3x3_raw')

    def rows_stats(self, Y, C, tiles, part):
        st = self.lib.mmmot_rows_stats(_ptr(Y), _ld(Y), C, _iptr(tiles.row0), _iptr(tiles.nrows), tiles.T, _ptr(part),
                                       self._stream())
        _lib.check(st, 'mmmot_rows_stats')

    def bn_relu_pool(self, Z, C, sc, sh, L, H, W, pool, A):
        st = self.lib.mmmot_bn_relu_pool(_ptr(Z), C, _ptr(sc), _ptr(sh), L, H, W, int(bool(pool)), _ptr(A), self._stream())
        _lib.check(st, 'mmmot_bn_relu_pool')

    def maxpool_bwd(self, Z, C, sc, sh, dP, L, H, W, dA):
        st = self.lib.mmmot_maxpool_bwd(_ptr(Z), C, _ptr(sc), _ptr(sh), _ptr(dP), L, H, W, _ptr(dA), self._stream())
        _lib.check(st, 'mmmot_maxpool_bwd')

    def conv3x3_wgrad(self, dZ, A, L, H, W, Cin, Cout, nsplit, dW, amax=None):
        if self.tn_f16:
            # fp16 matrix cores (3-term split), dZ scaled by a power of two taken from its maximum on the device
            # (`amax`: that maximum when the caller has it already - the input-gradient convolution uses the same one)
            if amax is None:
                amax = self._scratch(('tn_amax', int(self._stream() or 0)), 1, dZ.device)
                _lib.check(self.lib.mmmot_absmax(_ptr(dZ), Cout, L * H * W, Cout, _ptr(amax), self._stream()), 'mmmot_absmax')
            st = self.lib.mmmot_conv3x3_wgrad_f16(_ptr(dZ), _ptr(A), L, H, W, Cin, Cout, nsplit, _ptr(dW), _ptr(amax),
                                                  self._stream())
            _lib.check(st, 'mmmot_conv3x3_wgrad_f16')
            return
        st = self.lib.mmmot_conv3x3_wgrad(_ptr(dZ), _ptr(A), L, H, W, Cin, Cout, nsplit, _ptr(dW), self._stream())
        _lib.check(st, 'mmmot_conv3x3_wgrad')

    def conv3x3_first_wgrad(self, dZ, X, L, H, W, PW):
        st = self.lib.mmmot_conv3x3_first_wgrad(_ptr(dZ), _ptr(X), L, H, W, _ptr(PW), PW.shape[0], self._stream())
        _lib.check(st, 'mmmot_conv3x3_first_wgrad')

    def selftest_mfma(self, A, B, C, K):
        _lib.check(self.lib.mmmot_selftest_mfma(_ptr(A), _ptr(B), _ptr(C), K, self._stream()),
                   'mmmot_selftest_mfma')

"""Loader and builder of ``libmmmot_hip.so`` (the C-ABI of include/mmmot_hip.h).

The library is built in-tree with ``hipcc --offload-arch=gfx950`` (no torch
headers, no CMake) and loaded with ctypes.  There is NO fallback: if the shared
object is missing or fails to load, every op raises - the product path never
computes on the CPU.
"""
import ctypes
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIB_PATH = os.environ.get('MMMOT_LIB_PATH', os.path.join(_HERE, 'libmmmot_hip.so'))  # override: tools' timing-experiment builds
SOURCES = ['conv3x3.hip', 'hl16_format.hip', 'conv3x3_hl16_patch.hip', 'gemm_rows.hip', 'gemm_wide.hip',
           'gemm_ares.hip', 'gemm_wres.hip', 'gemm_wreg.hip', 'pn_mlp64.hip', 'gram.hip', 'points_gather.hip', 'crop_resize.hip', 'small_kernels.hip', 'backward.hip', 'train.hip', 'train_vgg.hip', 'gemm_tn_f16.hip']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
HIPFLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC']

_lock = threading.Lock()
_lib = None

c_f = ctypes.c_void_p  # device pointers travel as integers
c_i = ctypes.c_int


class GemmArgs(ctypes.Structure):
    """Mirror of ``mmmot_gemm_args`` (include/mmmot_hip.h)."""
    _fields_ = [
        ('X', c_f), ('ldx', c_i),
        ('W', c_f),
        ('bias', c_f),
        ('dbias', c_f), ('rowidx', c_f), ('lddb', c_i),
        ('Y', c_f), ('ldy', c_i),
        ('part', c_f),
        ('sc', c_f), ('sh', c_f), ('ldsc', c_i),
        ('FA', c_f), ('FB', c_f), ('ldf', c_i),
        ('tile_row0', c_f), ('tile_nrows', c_f), ('tile_group', c_f),
        ('grp_row0', c_f), ('grp_M', c_f), ('grp_aoff', c_f), ('grp_boff', c_f),
        ('T', c_i), ('N', c_i), ('K', c_i),
        ('amode', c_i), ('pairop', c_i), ('act', c_i),
        ('w_hl16', c_i), ('oscale', ctypes.c_float),
        ('osc', c_f), ('osh', c_f), ('ldosc', c_i),
        ('colsum', c_f),
        ('pair_uniform32', c_i),
    ]


class GemmAresArgs(ctypes.Structure):
    """Mirror of ``mmmot_gemm_ares_args`` (include/mmmot_hip.h)."""
    _fields_ = [
        ('X', c_f), ('ldx', c_i),
        ('sc', c_f), ('sh', c_f), ('ldsc', c_i),
        ('W', c_f),
        ('bias', c_f),
        ('dbias', c_f), ('tile_dbrow', c_f), ('lddb', c_i),
        ('tile_row0', c_f), ('tile_nrows', c_f), ('tile_group', c_f),
        ('part', c_f),
        ('osc', c_f), ('osh', c_f), ('ldosc', c_i),
        ('colsum', c_f),
        ('T', c_i), ('N', c_i), ('K', c_i),
        ('oscale', ctypes.c_float),
    ]


class GemmTnArgs(ctypes.Structure):
    """Mirror of ``mmmot_gemm_tn_args`` (include/mmmot_hip.h)."""
    _fields_ = [
        ('dY', c_f), ('lddy', c_i),
        ('X', c_f), ('ldx', c_i),
        ('sc', c_f), ('sh', c_f), ('ldsc', c_i),
        ('FA', c_f), ('FB', c_f), ('ldf', c_i),
        ('tile_row0', c_f), ('tile_nrows', c_f), ('tile_group', c_f),
        ('grp_row0', c_f), ('grp_M', c_f), ('grp_aoff', c_f), ('grp_boff', c_f),
        ('T', c_i), ('N', c_i), ('K', c_i), ('amode', c_i), ('pairop', c_i), ('nsplit', c_i),
        ('dW', c_f), ('db', c_f),
    ]


# name -> argtypes; every entry point declared in include/mmmot_hip.h
SIGNATURES = {
    'mmmot_abi_version': [],
    'mmmot_device_info': [c_i, ctypes.POINTER(c_i), ctypes.c_char_p, c_i],
    'mmmot_conv3x3_bn_relu': [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f],
    'mmmot_gemm_rows': [ctypes.POINTER(GemmArgs), c_f],
    'mmmot_set_gemm_rows_variant': [c_i],
    'mmmot_gemm_ares': [ctypes.POINTER(GemmAresArgs), c_f],
    'mmmot_set_gemm_ares_variant': [c_i],
    'mmmot_gram_rows': [c_f, c_i, c_i, c_f, c_f, c_i, c_f, c_f, c_f, c_i, c_f, c_f, c_f],
    'mmmot_set_gram128_variant': [c_i],
    'mmmot_gn_finalize_gram': [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_f, c_f, c_i, c_f, c_f, ctypes.c_float, c_f, c_f, c_f, c_f],
    # Gp, Sp, grp_tile0, grp_ntiles, grp_count, G, tile_nrows, tile_det, K, W, dbias, lddb, N, gamma, beta, eps, work, sc, sh, stream
    'mmmot_gn_finalize_gram_dbias': [c_f, c_f, c_f, c_f, c_f, c_i, c_f, c_f, c_i, c_f, c_f, c_i, c_i, c_f, c_f, ctypes.c_float,
                                     c_f, c_f, c_f, c_f],
    'mmmot_gn_finalize': [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f, c_f, ctypes.c_float, c_f, c_f, c_f],
    'mmmot_segment_mean': [c_f, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_i, c_f, c_f, c_i, c_i, c_f, c_i, c_i, c_f],
    'mmmot_conv3x3_bn_relu_hl16_patch': [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f],
    'mmmot_conv1_fused_hl16': [c_f, c_f, c_f, ctypes.c_float, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f],
    'mmmot_conv3x3_bn_relu_hq8': [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f],
    'mmmot_conv1_fused_u8': [c_f, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                             ctypes.c_float, c_f, c_f, ctypes.c_float, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f],
    'mmmot_u8_normalize': [c_f, c_i, c_i, c_f, c_f, c_f],
    'mmmot_conv1_fused_hq8': [c_f, c_f, c_f, ctypes.c_float, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f],
    'mmmot_conv3x3_first_hl16': [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f],
    'mmmot_set_patch_grid_limit': [c_i],
    'mmmot_set_patch_min_block': [c_i],
    'mmmot_trunk_range_read': [ctypes.POINTER(ctypes.c_uint), c_i],
    'mmmot_trunk_range_bind': [c_f],
    'mmmot_hl16_pack': [c_f, c_f, ctypes.c_long, c_f],
    'mmmot_hl16_unpack': [c_f, c_f, ctypes.c_long, c_f],
    'mmmot_hl16_pack_pow2': [c_f, c_f, ctypes.c_long, c_f, c_i, c_f],
    'mmmot_pow2_oscale': [c_f, c_i, c_f, c_i, c_f, c_i, c_f],
    'mmmot_conv3x3_raw_hl16': [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_f],
    'mmmot_hq8_pack': [c_f, c_f, ctypes.c_long, c_f],
    'mmmot_hq8_unpack': [c_f, c_f, ctypes.c_long, c_f],
    'mmmot_rowdot': [c_f, c_i, c_i, c_f, ctypes.c_float, c_f, c_f, c_i, c_f, c_f, c_f, c_i, c_i, c_i,
                     ctypes.c_float, c_f, c_f, c_f],
    'mmmot_row_layernorm': [c_f, c_i, c_i, c_f, c_f, ctypes.c_float, c_i, c_f, c_i, c_i, c_f],
    'mmmot_skippool_head': [c_f, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, ctypes.c_float, c_f, c_i, c_i,
                            c_f],
    'mmmot_pointnet_layer1': [c_f, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_f],
    'mmmot_pn_mlp64': [c_f, c_i, c_f, c_f, c_i, c_f, ctypes.c_float, c_f, c_f, c_i, c_f, c_f, c_f, c_f, c_i, c_i, c_f],
    'mmmot_affine_act': [c_f, c_i, c_i, c_f, c_f, c_i, c_f, c_f, c_f, c_i, c_i, c_f, c_i, c_f],
    'mmmot_fusion_combine': [c_i, c_f, c_f, c_i, c_f, c_i, c_f, c_f, c_f, c_f, c_i, c_f, c_f, c_f, c_i, c_f,
                             c_i, c_i, c_f],
    'mmmot_softmax_pairs': [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f],
    'mmmot_points_count': [c_f, c_i, c_i, c_f, c_i, c_i, c_f, c_f, c_f],
    'mmmot_points_scatter': [c_f, c_i, c_i, c_f, c_i, c_f, c_f, c_f, c_i, c_f],
    'mmmot_crop_resize_norm': [c_f, c_i, c_i, c_f, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_f],
    'mmmot_points_count_batched': [c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_f, c_f, c_f],
    'mmmot_points_scatter_batched': [c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_f],
    'mmmot_selftest_mfma': [c_f, c_f, c_f, c_i, c_f],
    'mmmot_gn_bwd_partial': [c_f, c_i, c_f, c_i, c_i, c_f, c_f, c_i, c_f, c_f, c_i, c_f, c_f, c_f, c_i, c_f, c_f],
    'mmmot_gn_bwd_finalize': [c_f, c_f, c_i, c_i, c_i, c_f, c_f, c_f],
    'mmmot_gn_bwd_apply': [c_f, c_i, c_f, c_i, c_i, c_f, c_f, c_i, c_f, c_f, c_i, c_f, c_f, c_f, c_f, c_i, c_f, c_i, c_f],
    'mmmot_gemm_tn': [ctypes.POINTER(GemmTnArgs), c_f],
    'mmmot_gemm_tn_f16': [ctypes.POINTER(GemmTnArgs), c_f, c_f],
    'mmmot_absmax': [c_f, c_i, ctypes.c_long, c_i, c_f, c_f],
    'mmmot_pair_bwd': [c_f, c_i, c_f, c_i, c_f, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f],
    'mmmot_pair_expand_bwd': [c_f, c_i, c_f, c_i, c_i, c_f, c_f, c_f, c_i, c_f, c_f, c_f, c_f, c_f],
    'mmmot_rowdot_bwd': [c_f, c_i, c_i, c_f, ctypes.c_float, c_f, c_f, c_i, c_f, c_f, c_f, c_i, c_i, c_f, c_f, c_f, c_i,
                         c_f, c_i, c_f],
    'mmmot_softmax_pairs_bwd': [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f],
    'mmmot_fusion_c_bwd': [c_f, c_f, c_i, c_f, c_i, c_f, c_f, c_f, c_f, c_i, c_f, c_f, c_f, c_i, c_f, c_f, c_i, c_f, c_f, c_i, c_f],
    'mmmot_add_rows': [c_f, c_i, c_f, c_i, c_f, c_i, ctypes.c_long, c_i, c_f],
    'mmmot_conv3x3_raw': [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_f],
    'mmmot_rows_stats': [c_f, c_i, c_i, c_f, c_f, c_i, c_f, c_f],
    'mmmot_bn_relu_pool': [c_f, c_i, c_f, c_f, c_i, c_i, c_i, c_i, c_f, c_f],
    'mmmot_maxpool_bwd': [c_f, c_i, c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_f],
    'mmmot_conv3x3_wgrad': [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f],
    'mmmot_conv3x3_wgrad_f16': [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f],
    'mmmot_conv3x3_first_wgrad': [c_f, c_f, c_i, c_i, c_i, c_f, c_i, c_f],
    'mmmot_rows_gather_scale': [c_f, c_i, c_f, c_f, c_f, c_i, ctypes.c_long, c_i, c_f],
    'mmmot_pointnet_layer1_bwd': [c_f, c_f, c_i, c_f, c_f, c_i, c_f, c_f],
    'mmmot_score_loss': [c_f, c_i, c_f, c_f, c_f, c_i, c_i, ctypes.c_float, c_i, ctypes.c_float, c_i, c_i, c_f, c_i, c_f,
                         c_i, c_i, c_f],
    'mmmot_ghm_loss': [c_f, c_i, c_f, ctypes.c_float, ctypes.c_float, c_i, c_i, c_i, ctypes.c_float, c_f, c_f, c_i, c_f, c_i,
                       c_f],
}


# entry points that exist only in -DMMMOT_DEBUG builds (timing experiments of tools/; never the product library)
DEBUG_SIGNATURES = {
    'mmmot_set_patch_variant': [c_i],
    'mmmot_debug_read_patch_timers': [ctypes.POINTER(ctypes.c_ulonglong), c_i],
}
DEBUG_LIB_PATH = os.path.join(_HERE, 'libmmmot_hip_debug.so')


def build(force=False, verbose=False, debug=False):
    """Compile every HIP source for gfx950 and link libmmmot_hip.so in-tree.  ``debug=True`` builds the
    -DMMMOT_DEBUG variant (timing experiments of the patch kernel) as libmmmot_hip_debug.so; load it by setting
    MMMOT_LIB_PATH before importing mmmot_amd (tools/ do)."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    # headers and the textual parts of the trunk kernel (csrc/patch_*.inc): any of them newer than the library rebuilds it
    deps = srcs + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(('.h', '.inc'))] + \
        [os.path.join(_HERE, '..', 'include', 'mmmot_hip.h')]
    lib_path = DEBUG_LIB_PATH if debug else LIB_PATH
    if not force and os.path.exists(lib_path):
        if all(os.path.getmtime(lib_path) >= os.path.getmtime(d) for d in deps):
            return lib_path
    objs = []
    procs = []
    for s in srcs:
        o = s[:-4] + ('.dbg.o' if debug else '.o')
        objs.append(o)
        cmd = [HIPCC] + HIPFLAGS + (['-DMMMOT_DEBUG'] if debug else []) + ['-c', s, '-o', o]
        if verbose:
            print(' '.join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError('hipcc failed: %s\n%s' % (' '.join(cmd), out.decode()))
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib_path] + objs
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return lib_path


def load():
    """Return the ctypes handle; raises (never falls back) when unavailable."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    'libmmmot_hip.so is not built (%s). Run `python -c "import __graft_entry__ as g; '
                    'g.build()"`. There is no CPU fallback for the HIP path.' % LIB_PATH)
            lib = ctypes.CDLL(LIB_PATH)
            # MMMOT_LIB_ALLOW_MISSING=1 (tools/ab_forward.py only: an OLDER build of the library beside the current one)
            # tolerates entry points the other build does not have yet; the product never sets it
            lenient = os.environ.get('MMMOT_LIB_ALLOW_MISSING', '0') == '1'
            for name, argtypes in SIGNATURES.items():
                fn = getattr(lib, name, None) if lenient else getattr(lib, name)  # AttributeError if the symbol is missing
                if fn is None:
                    continue
                fn.argtypes = argtypes
                fn.restype = c_i
            for name, argtypes in DEBUG_SIGNATURES.items():  # present in -DMMMOT_DEBUG builds only
                fn = getattr(lib, name, None)
                if fn is not None:
                    fn.argtypes = argtypes
                    fn.restype = c_i
            _lib = lib
    return _lib


def check(status, what):
    if status != 0:
        raise RuntimeError('%s failed with status %d%s' % (
            what, status, ' (MMMOT_EINVAL: contract violation)' if status == -1 else ' (hipError_t)'))

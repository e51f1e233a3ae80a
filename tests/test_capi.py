"""The C-ABI shared library builds for gfx950, loads without a GPU and exports
every symbol that include/mmmot_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

from mmmot_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(debug=False):
    """entry points of include/mmmot_hip.h; the `#ifdef MMMOT_DEBUG` block (timing experiments of tools/, absent from
    the product library) is returned separately with debug=True"""
    text = open(os.path.join(ROOT, 'include', 'mmmot_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    dbg = ''.join(re.findall(r'#ifdef MMMOT_DEBUG(.*?)#endif', text, flags=re.S))
    if not debug:
        text = re.sub(r'#ifdef MMMOT_DEBUG.*?#endif', '', text, flags=re.S)
    else:
        text = dbg
    return sorted(set(re.findall(r'\bint\s+(mmmot_\w+)\s*\(', text)))


def test_library_builds_and_exports_every_declared_symbol():
    path = _lib.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    syms = declared_symbols()
    assert len(syms) >= 13, syms
    for s in syms:
        assert hasattr(lib, s), 'symbol %s declared in include/mmmot_hip.h but not exported' % s
    assert set(syms) == set(_lib.SIGNATURES), (set(syms) ^ set(_lib.SIGNATURES))
    # experiment knobs live behind -DMMMOT_DEBUG: declared, bound by _lib when present, NOT in the product library
    dbg = declared_symbols(debug=True)
    assert set(dbg) == set(_lib.DEBUG_SIGNATURES) and dbg
    for s in dbg:
        assert not hasattr(lib, s), 'debug entry point %s leaked into the product library' % s


def test_abi_version_and_argument_checks_without_gpu():
    lib = _lib.load()
    assert lib.mmmot_abi_version() == 10
    # contract violations are rejected before any launch (safe on a GPU-less host)
    assert lib.mmmot_gemm_rows(None, None) == -1
    assert lib.mmmot_conv3x3_bn_relu(None, None, None, None, 1, 8, 8, 64, 64, 0, 0, None) == -1
    assert lib.mmmot_softmax_pairs(None, None, None, None, None, 1, 4, 3, None) == -1
    # mmmot_pn_mlp64: null pointers, and (with non-null, 16-byte aligned dummies) an output width other than 64 / 128
    assert lib.mmmot_pn_mlp64(None, 64, None, None, 64, None, 1.0, None, None, 64, None, None, None, None, 1, 64, None) == -1
    d = 4096  # never dereferenced: the argument checks come before any launch
    assert lib.mmmot_pn_mlp64(d, 64, d, d, 64, d, 1.0, d, d, 96, d, d, d, None, 1, 96, None) == -1
    assert lib.mmmot_pn_mlp64(d, 62, d, d, 64, d, 1.0, d, d, 64, d, d, d, None, 1, 64, None) == -1


import pytest


@pytest.mark.parametrize('cname,mirror', [('mmmot_gemm_args', 'GemmArgs'), ('mmmot_gemm_ares_args', 'GemmAresArgs'),
                                          ('mmmot_gemm_tn_args', 'GemmTnArgs')])
def test_args_struct_layouts_match_header(tmp_path, cname, mirror):
    """The ctypes mirrors must have the C structs' sizes and field offsets: compile the header with gcc
    (which also proves include/mmmot_hip.h is plain C) and compare."""
    import subprocess
    cls = getattr(_lib, mirror)
    fields = [f[0] for f in cls._fields_]
    src = tmp_path / 'layout.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mmmot_hip.h"\nint main(void){\n'
                   'printf("%%zu\\n", sizeof(%s));\n' % cname +
                   ''.join('printf("%%zu\\n", offsetof(%s, %s));\n' % (cname, f) for f in fields) +
                   'return 0;}\n')
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-std=c99', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    nums = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert nums[0] == ctypes.sizeof(cls)
    assert nums[1:] == [getattr(cls, f).offset for f in fields]

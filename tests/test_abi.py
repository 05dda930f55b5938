"""The C-ABI library loads and exports every symbol declared in include/xas_hip.h, and the ctypes
signature table matches the header.  No compute calls (CPU-only test)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_prototypes():
    src = open(os.path.join(ROOT, 'include', 'xas_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    src = re.sub(r'//[^\n]*', '', src)
    protos = {}
    for m in re.finditer(r'\b(int|size_t|const char\*)\s+(xas_\w+)\s*\(([^;{]*?)\)\s*;', src, flags=re.S):
        ret, name, args = m.groups()
        codes = ''
        args = args.strip()
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                if 'xas_conv_shape' in a:
                    codes += 's'
                elif '*' in a:
                    codes += 'p'
                elif re.match(r'(const\s+)?unsigned\b', a):
                    codes += 'u'
                elif re.match(r'(const\s+)?long\b', a):
                    codes += 'l'
                elif re.match(r'(const\s+)?int\b', a):
                    codes += 'i'
                elif re.match(r'(const\s+)?float\b', a):
                    codes += 'f'
                elif re.match(r'(const\s+)?double\b', a):
                    codes += 'd'
                elif re.match(r'(const\s+)?size_t\b', a):
                    codes += 'z'
                else:
                    raise AssertionError('unparsed argument %r in %s' % (a, name))
        protos[name] = (codes, {'int': 'i', 'size_t': 'z', 'const char*': 'c'}[ret])
    return protos


def test_library_is_built():
    import __graft_entry__ as ge
    if not os.path.exists(ge.LIB):
        ge.build_lib(verbose=False)
    assert os.path.exists(ge.LIB)


def test_exports_and_signatures():
    import ctypes
    from xas_amd import _lib
    lib = _lib.load()
    protos = header_prototypes()
    assert len(protos) > 30
    for name, (codes, ret) in protos.items():
        assert hasattr(lib, name), 'library does not export %s' % name
        if name == 'xas_last_error':
            continue
        assert name in _lib.SIGNATURES, 'no ctypes signature for %s' % name
        assert _lib.SIGNATURES[name] == (codes, ret), '%s: table %r vs header %r' % (name, _lib.SIGNATURES[name], (codes, ret))
    assert set(_lib.SIGNATURES) <= set(protos)
    assert ctypes.sizeof(_lib.ConvShape) == 12 * 4 + 16 and _lib.ConvShape.grad_amax.offset == 48 and _lib.ConvShape.x_amax.offset == 56      # 12 ints + the two operand-maximum pointers
    assert lib.xas_abi_version() == 3
    # recorded maxima: the slot layout of the header is the one the Python side allocates
    import re
    from xas_amd import ops_nn
    src = open(os.path.join(ROOT, 'include', 'xas_hip.h')).read()
    sub, stride = (int(re.search(r'#define %s (\d+)' % n, src).group(1)) for n in ('XAS_AMAX_SUB', 'XAS_AMAX_STRIDE'))
    assert sub * stride == ops_nn.AMAX_SLOT_FLOATS and stride * 4 == 128


def test_no_cpu_fallback():
    import torch
    from xas_amd import ops_head
    with pytest.raises(RuntimeError):
        ops_head.softargmax_multi(torch.zeros(1, 32, 16, 16), 2, 3, 15)


def test_argument_checks_return_errors_without_a_gpu():
    """Every entry point validates its arguments BEFORE touching the device: status 1 and a message, no launch."""
    import ctypes
    from xas_amd import _lib
    lib = _lib.load()
    one = ctypes.c_void_p(16)                       # any non-null pointer: never dereferenced on these paths
    err = lambda: lib.xas_last_error().decode()
    f = _lib.fn
    assert f('xas_eval_select')(one, one, one, 2, 3, 65, 3, 256.0, 0, None, None, None, None, None) == 1
    assert 'K <= 64' in err()
    assert f('xas_eval_select')(one, one, one, 2, 3, 18, 4, 256.0, 0, None, None, None, None, None) == 1
    assert f('xas_triangulate_dlt')(one, one, 2, 1, 18, one, None) == 1 and '2 views' in err()
    assert f('xas_pose_metrics')(one, one, None, 2, 2, 1.0, 0, 0.15, None, None, None, None, None) == 1
    assert f('xas_pose_metrics')(one, one, None, 2, 18, 1.0, 3, 0.15, None, None, None, None, None) == 1
    assert f('xas_projection_matrix')(None, one, one, 2, one, None) == 1
    shp = _lib.ConvShape(2, 8, 8, 3, 64, 3, 3, 1, 1, 8, 8)         # Cin = 3: not on the accumulating MFMA path
    assert f('xas_conv_dgrad_acc')(one, one, one, ctypes.byref(shp), None) == 1 and 'MFMA path' in err()
    assert f('xas_bn_bwd_reduce')(None, None, one, one, one, None, None, 1e-5, 1, 64, 64, 1, one, one, None, None, None, None) == 1
    assert f('xas_bn_bwd_apply')(None, one, one, one, one, one, None, one, 1e-5, 1, 64, 64, 1, 64.0, one, None, None, None) == 1
    assert 'invertible' in err().lower() or 'null buffer' in err()

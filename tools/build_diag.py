#!/usr/bin/env python3
"""Builds x-as-supervision_amd/xas_amd/libxas_hip_diag.so: the product library with conv.hip compiled under
-DXAS_CONV_DIAG (in-kernel phase stamps + ablation flags).  Use it with XAS_HIP_LIB=<that path> python tools/stamp_conv.py."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

g.build_lib(verbose=False)
obj = os.path.join(g.PKG, 'build', 'conv_diag.o')
subprocess.check_call([g.HIPCC] + g.FLAGS + ['-DXAS_CONV_DIAG', '-c', os.path.join(g.CSRC, 'conv.hip'), '-o', obj])
objs = [os.path.join(g.PKG, 'build', f) for f in sorted(os.listdir(os.path.join(g.PKG, 'build')))
        if f.endswith('.o') and f not in ('conv.o', 'conv_diag.o')] + [obj]
out = os.path.join(g.PKG, 'xas_amd', 'libxas_hip_diag.so')
subprocess.check_call([g.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs)
print(out)

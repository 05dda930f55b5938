#!/usr/bin/env python3
"""Ablation builds of the bf16x6 igemm kernel: one library per XAS_X6_ABL mask under x-as-supervision_amd/xas_amd/abl/
(git-ignored, travels to the GPU box).  usage: python tools/build_abl.py 1 2 4 ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

g.build_lib(verbose=False)
out = os.path.join(g.PKG, 'xas_amd', 'abl')
os.makedirs(out, exist_ok=True)
objdir = os.path.join(g.PKG, 'build')
others = [os.path.join(objdir, f) for f in os.listdir(objdir) if f.endswith('.o') and f != 'conv_x6.o' and not f.startswith('conv_x6_abl')]
procs = []
extra = os.environ.get('XAS_ABL_DEFS', '').split()      # e.g. -DXAS_X6_LDS_FLOOR=61440: two blocks per CU for every variant
for m in sys.argv[1:]:
    o = os.path.join(objdir, 'conv_x6_abl%s.o' % m)
    procs.append((m, o, subprocess.Popen([g.HIPCC] + g.FLAGS + extra + ['-DXAS_X6_ABL=' + m.split('o')[0], '-c', os.path.join(g.CSRC, 'conv_x6.hip'), '-o', o])))
for m, o, p in procs:
    if p.wait() != 0:
        raise SystemExit('compile failed for mask ' + m)
    subprocess.check_call([g.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(out, 'libxas_abl%s.so' % m), o] + others)
    print('built', m)

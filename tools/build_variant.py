#!/usr/bin/env python3
"""Build a variant of the whole library with extra -D flags into x-as-supervision_amd/xas_amd/abl/libxas_<name>.so
(for in-box A/Bs with tools/gpu/ab_lib.sh).  usage: python tools/build_variant.py <name> -DFOO=1 [-DBAR=2 ...]"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

name, defs = sys.argv[1], sys.argv[2:]
objdir = os.path.join(g.PKG, 'build', 'var_' + name)
os.makedirs(objdir, exist_ok=True)
procs, objs = [], []
for src in sorted(glob.glob(os.path.join(g.CSRC, '*.hip'))):
    o = os.path.join(objdir, os.path.basename(src)[:-4] + '.o')
    objs.append(o)
    procs.append(subprocess.Popen([g.HIPCC] + g.FLAGS + defs + ['-c', src, '-o', o]))
for p in procs:
    if p.wait() != 0:
        raise SystemExit('compile failed')
out = os.path.join(g.PKG, 'xas_amd', 'abl')
os.makedirs(out, exist_ok=True)
subprocess.check_call([g.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(out, 'libxas_%s.so' % name)] + objs)
print('built', name)

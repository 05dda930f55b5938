# A/B of a compile-time switch on ONE box: bash tools/gpu/ab_build.sh "-DXAS_NT_LOADS"   (base, defs, base)
python tools/ab_step.py 0 2>/dev/null | head -1 | sed "s/^/base   /"
XAS_HIPCC_DEFS="$1" python -c "import __graft_entry__ as g; g.build_lib(force=True, verbose=False)" > /dev/null 2>&1
python tools/ab_step.py 0 2>/dev/null | head -1 | sed "s/^/defs   /"
python -c "import __graft_entry__ as g; g.build_lib(force=True, verbose=False)" > /dev/null 2>&1
python tools/ab_step.py 0 2>/dev/null | head -1 | sed "s/^/base   /"

python tools/bench_ops.py prec 2>/dev/null | sed "s/^/base  /"
XAS_HIPCC_DEFS="$1" python -c "import __graft_entry__ as g; g.build_lib(force=True, verbose=False)" > /dev/null 2>&1
python tools/bench_ops.py prec 2>/dev/null | sed "s/^/defs  /"

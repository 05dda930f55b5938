mkdir -p gpurun_out/r02d
python -m pytest tests -m gpu -q -x > gpurun_out/r02d/gputests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r02d/gputests.log
tail -5 gpurun_out/r02d/gputests.log
python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r02d/bench.json 2> gpurun_out/r02d/bench.err
tail -3 gpurun_out/r02d/bench.err
XAS_CAM_BATCH=0 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r02d/bench_nobatch.json 2> gpurun_out/r02d/bench_nobatch.err
tail -2 gpurun_out/r02d/bench_nobatch.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02d/prof -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r02d/prof.log 2>&1

set -x
mkdir -p gpurun_out/r02a
python -m pytest tests -m gpu -x -q > gpurun_out/r02a/gputests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r02a/gputests.log
python bench.py --steps 5 --warmup 2 > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02a/prof -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r02a/prof.log 2>&1
ls -la $GRAFT_REPO_ROOT/gpurun_out/r02a/prof

#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run() { echo "#### $*"; env "$@" timeout -k 10 500 python tools/diag_repro.py --workload HM36_Multi_SurS2 --bisect --only-default --no-poison --loops 50 2>&1 | grep -a "==\|DIFFERS\|tap \|\[b " | cut -c1-200; }
run GPU_MAX_HW_QUEUES=2
run GPU_MAX_HW_QUEUES=8
run XAS_NOOP=1

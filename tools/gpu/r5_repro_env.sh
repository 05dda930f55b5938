#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run() { echo "#### $*"; env "$@" timeout -k 10 500 python tools/diag_repro.py --workload HM36_Multi_SurS2 --bisect --only-default --no-poison --loops 50 2>&1 | grep -a "==\|DIFFERS" | cut -c1-120; }
run XAS_AUX_IS_SIDE=1
run XAS_ADV_AUX=0
run HSA_ENABLE_SDMA=0

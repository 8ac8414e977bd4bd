#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_any.sh "<counters>" <kernel name pattern> <bench args...>
# one counter pass (kernel trace only) over a short bench run; prints the per-launch means of the matching kernels
set -e -o pipefail
ctrs=$1; pat=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pmc_any
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pmc_any -- python3 bench.py \
    --steps 5 --warmup 1 --no-cpu-baseline --no-other-workloads "$@" > /tmp/pmc_any.log 2>&1
python3 tools/pmc_summary.py /tmp/pmc_any | grep -A9 -E "$pat"

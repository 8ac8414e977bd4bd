#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_one.sh <kernel name pattern> <bench args...>
# one SQ counter pass over a short bench run; prints the per-launch means of the matching kernels
set -e -o pipefail
pat=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pmc_one
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS \
    SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU --output-format csv -d /tmp/pmc_one -- python3 bench.py \
    --steps 5 --warmup 1 --no-cpu-baseline --no-other-workloads "$@" > /tmp/pmc_one.log 2>&1
python3 tools/pmc_summary.py /tmp/pmc_one | grep -A9 -E "$pat"

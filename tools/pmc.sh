#!/bin/bash
# usage: tools/pmc.sh <outdir> <counters (space separated)> -- <python args...>
# One rocprofv3 --pmc pass with --kernel-trace only, as the pool's rules require.
set -e
out=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d "$out" -- python3 "$@"

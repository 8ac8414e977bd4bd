#!/bin/bash
# usage: tools/gp.sh <timeout-s> '<command>'   -- gpurun with the scratch directory in place
t=$1; shift
exec /usr/local/graft/bin/gpurun --timeout "$t" -- "mkdir -p gpurun_out/r3; $*"

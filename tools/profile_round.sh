#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>
# 1. rocprofv3 --kernel-trace --stats over the default bench.py run
# 2. two --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel trace only) over a short bench run
# 3. summaries into gpurun_out/profiles_<tag>/ (copy them to profiles/ to commit)
set -e -o pipefail
tag=$1
out=gpurun_out/profiles_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 bench.py \
    > "$out/bench_under_rocprof.json" 2> /tmp/prof_stats.err
stats=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1)
head -1 "$stats" > "$out/kernel_stats_iak.csv"
grep "iak::" "$stats" >> "$out/kernel_stats_iak.csv"
cp "$stats" "$out/kernel_stats_all.csv"
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -- python3 bench.py \
    --steps 5 --warmup 1 --no-cpu-baseline > /tmp/pmc_fetch.log 2>&1
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -- python3 bench.py \
    --steps 5 --warmup 1 --no-cpu-baseline > /tmp/pmc_write.log 2>&1
python3 tools/make_traffic.py /tmp/pmc_fetch /tmp/pmc_write 500 "$out/traffic.json" > /dev/null
# the plain run, with the measured traffic in place, is the line to compare with the profile
mkdir -p profiles && cp "$out/traffic.json" profiles/traffic.json
timeout -k 10 300 python3 bench.py > "$out/bench.json"
cat "$out/bench.json"

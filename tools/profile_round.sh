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
# SQ counters: instruction-issue time, LDS activity and conflicts per kernel of the bench ...
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS \
    SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d /tmp/pmc_sq -- python3 bench.py \
    --steps 5 --warmup 1 --no-cpu-baseline > /tmp/pmc_sq.log 2>&1
{
    echo "# mean per launch, bench.py batch (500 x 768x568); SQ_* cycle counters are in units of 4 clocks, summed over the chip"
    python3 tools/pmc_summary.py /tmp/pmc_sq
} > "$out/pmc_sq_bench.txt"
# ... and the histogram's LDS atomics on the three input distributions (cell-like, constant, uniform)
for kind in C K U; do
    timeout -s KILL 150 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT \
        SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d /tmp/pmc_hist_$kind -- python3 tools/kbench.py \
        --only gray_hist,hist --reps 5 --kind $kind > /tmp/pmc_hist_$kind.log 2>&1
    {
        echo "# kind $kind: $(grep -E '^(gray_hist|hist) ' /tmp/pmc_hist_$kind.log | tr -s ' ' | tr '\n' ';')"
        python3 tools/pmc_summary.py /tmp/pmc_hist_$kind | grep -A8 -E "gray_kernel<0, true>|hist_kernel"
    } >> "$out/pmc_sq_histogram.txt"
done
python3 tools/make_traffic.py /tmp/pmc_fetch /tmp/pmc_write 500 "$out/traffic.json" > /dev/null
# the plain run, with the measured traffic in place, is the line to compare with the profile
mkdir -p profiles && cp "$out/traffic.json" profiles/traffic.json
timeout -k 10 300 python3 bench.py > "$out/bench.json"
cat "$out/bench.json"

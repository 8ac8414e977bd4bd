#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>
# 1. rocprofv3 --kernel-trace --stats over the bench (P-A alone, then P-B and P-C alone)
# 2. --pmc passes (kernel trace only): FETCH_SIZE, WRITE_SIZE and the SQ counters over P-A, the SQ counters over P-B and P-C
# 3. summaries into gpurun_out/profiles_<tag>/ (copy them to profiles/ to commit)
set -e -o pipefail
tag=$1
out=gpurun_out/profiles_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
SQ="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS"
stats_of() {  # <name> <bench args...>: kernel stats of one bench run
    local name=$1; shift
    rm -rf /tmp/prof_$name
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python3 bench.py "$@" \
        > "$out/bench_under_rocprof_$name.json" 2> /tmp/prof_$name.err
    local stats; stats=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
    head -1 "$stats" > "$out/kernel_stats_$name.csv"
    grep "iak::" "$stats" >> "$out/kernel_stats_$name.csv"
    echo "stats $name done"
}
pmc_of() {  # <dir> <counters> -- <bench args...>
    local d=$1; shift
    local ctrs=()
    while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
    shift
    rm -rf "$d"
    timeout -s KILL 150 rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d "$d" -- python3 bench.py \
        --steps 5 --warmup 1 --no-cpu-baseline --no-other-workloads "$@" > "$d.log" 2>&1
    echo "pmc $d done"
}
stats_of PA --no-other-workloads
# config 3's per-GPU share (1250 images): the intermediates (545 MB each) no longer fit the 256 MiB Infinity Cache
stats_of PA_B1250 --no-other-workloads --batch 1250 --no-cpu-baseline
stats_of PB --workload P-B --steps 20 --no-cpu-baseline
stats_of PC --workload P-C --no-cpu-baseline
pmc_of /tmp/pmc_fetch FETCH_SIZE --
pmc_of /tmp/pmc_write WRITE_SIZE --
pmc_of /tmp/pmc_sq_PA $SQ --
pmc_of /tmp/pmc_fetch_B1250 FETCH_SIZE -- --batch 1250
pmc_of /tmp/pmc_write_B1250 WRITE_SIZE -- --batch 1250
pmc_of /tmp/pmc_sq_PAcfg $SQ -- --linear-kernel config
pmc_of /tmp/pmc_sq_PB $SQ -- --workload P-B
pmc_of /tmp/pmc_sq_PC $SQ -- --workload P-C
for w in PA PB PC; do
    {
        echo "# mean per launch over the bench's default batch of workload $w; SQ_* cycle counters are in units of 4 clocks, summed over the chip"
        python3 tools/pmc_summary.py /tmp/pmc_sq_$w
    } > "$out/pmc_sq_$w.txt"
done
python3 tools/make_valu.py "$out/valu.json" P-A=/tmp/pmc_sq_PA P-A/config=/tmp/pmc_sq_PAcfg P-B=/tmp/pmc_sq_PB P-C=/tmp/pmc_sq_PC
# the histogram's LDS atomics on the three input distributions (cell-like, constant, uniform)
for kind in C K U; do
    rm -rf /tmp/pmc_hist_$kind
    timeout -s KILL 150 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT \
        SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d /tmp/pmc_hist_$kind -- python3 tools/kbench.py \
        --only gray_hist,hist --reps 5 --kind $kind > /tmp/pmc_hist_$kind.log 2>&1
    {
        echo "# kind $kind (times under counter collection are inflated): $(grep -E '^(gray_hist|hist) ' /tmp/pmc_hist_$kind.log | tr -s ' ' | tr '\n' ';')"
        python3 tools/pmc_summary.py /tmp/pmc_hist_$kind | grep -A8 -E "gray_kernel<0, true>|hist_kernel"
    } >> "$out/pmc_sq_histogram.txt"
done
# ... and the same entry points timed plainly (no counters): the figures to quote
timeout -k 10 120 python3 tools/kbench.py --only gray_hist,hist,gray --reps 20 > "$out/kbench_hist.txt" 2>&1 || true
python3 tools/make_traffic.py /tmp/pmc_fetch /tmp/pmc_write 500 "$out/traffic.json" > /dev/null
python3 tools/make_traffic.py /tmp/pmc_fetch_B1250 /tmp/pmc_write_B1250 1250 "$out/traffic_B1250.json" > /dev/null
# the plain run, with the measured traffic and issue times in place, is the line to compare with the profiles
mkdir -p profiles && cp "$out/traffic.json" profiles/traffic.json && cp "$out/traffic_B1250.json" profiles/traffic_B1250.json && cp "$out/valu.json" profiles/valu.json
timeout -k 10 300 python3 bench.py > "$out/bench.json"
timeout -k 10 300 python3 bench.py --batch 1250 --no-cpu-baseline > "$out/bench_B1250.json"
timeout -k 10 300 python3 bench.py --batch 2000 --no-cpu-baseline > "$out/bench_B2000.json"
cat "$out/bench.json"

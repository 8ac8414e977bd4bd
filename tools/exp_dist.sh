set -e
for flags in "" "--force-dist --reduce-every 1" "--force-dist --reduce-every 8" "--force-dist --reduce-every 32" "--force-dist --reduce-every 8 --workload P-C" "--workload P-C"; do
  echo "== flags: $flags" >> gpurun_out/dist_overhead.txt
  timeout -k 10 150 python bench.py --no-cpu-baseline --no-other-workloads --steps 200 $flags 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read())
print(l['ms_per_step'], l['dataset'], l['config']['parallelism'])" >> gpurun_out/dist_overhead.txt
done

timeout -k 10 400 python -u -m pytest tests/test_gpu_ops.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -m gpu -x -q -k "median or fuzz or config4 or fullsize" --timeout 120 --timeout-method thread 2>&1 | tail -2 || exit 1
timeout -k 10 100 python bench.py --workload P-B --steps 10 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], [ (g['name'], g['ms']) for g in d['groups']])"
timeout -k 10 100 python tools/kbench.py --only median7,median5 --reps 10 2>/dev/null | grep "^median"

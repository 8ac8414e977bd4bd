cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pk
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -- python3 bench.py --workload P-C --steps 30 --no-cpu-baseline --no-other-workloads > /dev/null 2>&1 || exit 1
python3 tools/kstats.py /tmp/pk
timeout -k 10 100 python bench.py --workload P-C --steps 30 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], [ (g['name'], g['ms']) for g in d['groups']])"
timeout -k 10 100 python bench.py --steps 50 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], [ (g['name'], g['ms']) for g in d['groups']])"

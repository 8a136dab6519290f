cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c5p
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', '%.4f ms/launch' % r['avg_launch_ms'], '%.1f M agent-steps/s' % (d['value']/1e6), r.get('store_roof_GBps_this_box'))" >> gpurun_out/c5p/ab.txt; }
timeout 200 python bench.py --workload C5 --no-cpu-baseline --no-workloads --steps 60 --warmup 6 2>/dev/null | line "C5 60/6"
timeout 200 python bench.py --workload C5 --no-cpu-baseline --no-workloads --steps 100 --warmup 10 2>/dev/null | line "C5 100/10"
timeout 200 python bench.py --workload C5 --no-cpu-baseline --no-workloads --steps 300 --warmup 30 2>/dev/null | line "C5 300/30"
timeout 200 python bench.py --workload C5 --no-cpu-baseline --no-workloads --steps 20 --warmup 60 2>/dev/null | line "C5 20/60"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null > gpurun_out/c5p/default.json
python - <<'P' >> gpurun_out/c5p/ab.txt
import json
d=json.loads(open('gpurun_out/c5p/default.json').read().strip().splitlines()[-1])
print('default C2', d['value']/1e6, d['ms_per_step'])
for k,v in d['workloads'].items():
    print('default side', k, v.get('value',0)/1e6, v.get('ms_per_step'), v.get('roofline',{}).get('avg_launch_ms'), v.get('error'))
P
cat gpurun_out/c5p/ab.txt

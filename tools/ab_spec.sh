for m in 0 32768; do
  echo "=== skipmask=$m"
  AIE_DEV_SKIP_MASK=$m python bench.py --no-cpu-baseline --steps 1000 --warmup 100 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), round(d['ms_per_step']*1e3,1), round(d['roofline']['avg_launch_ms']*1e3,2))"
  AIE_DEV_SKIP_MASK=$m timeout 200 python tools/block_trace.py 2>&1 | tail -26 | head -14
done

for pad in 0 3000 6000 10000 16000 24000; do
  for w in 2 1; do
    echo -n "pad=$pad waves=$w: "
    AIE_DEV_LDS_PAD=$pad AIE_DEV_STEP_WAVES=$w python bench.py --no-cpu-baseline --steps 600 --warmup 100 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), round(d['ms_per_step']*1e3,1), round(d['roofline']['avg_launch_ms']*1e3,1))"
  done
done

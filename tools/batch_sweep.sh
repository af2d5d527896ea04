for b in 16384 32768; do
  s=$(date +%s)
  python bench.py --batch $b --steps 100 --warmup 10 --no-cpu --latency-ticks 50 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['batch_per_gpu'], d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
  echo "wall $(( $(date +%s) - s )) s"
done

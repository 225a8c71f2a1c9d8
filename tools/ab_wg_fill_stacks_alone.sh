for rep in 1 2 3; do for v in 0 1; do
  CRK_WG_FILL=$v timeout 300 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('CRK_WG_FILL=$v rep=$rep ms_per_step=%.4f stacks_alone_ms=%.4f roofline_frac=%.4f' % (d['ms_per_step'], d['stacks_alone']['ms'], d['roofline']['frac']))"
done; done

#!/bin/bash
# kernel-time ablation of the fused residual-block forward (CRK_DBG bits, see conv_kernels.hip)
for d in 0 1024 2048 4096 8192 15360; do
  CRK_DBG=$d python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
c=d['roofline']['classes']
print('dbg=$d', 'ms/step %.2f' % d['ms_per_step'], ' '.join('%s=%.1fus' % (k.split('<')[1][:6] if '<' in k else 'wgrad', v['avg_us']) for k,v in c.items()))
"
done

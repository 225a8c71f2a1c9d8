#!/bin/bash
# Runs the RCCL world-of-one graph-chain worker of tests/test_gpu_dp.py N times and reports how each run ended.
#   tools/rccl_loop.sh N OUTDIR [ENV=VALUE ...]
n=$1; out=$2; shift 2
mkdir -p "$out"
ok=0; bad=0
for i in $(seq 1 "$n"); do
  port=$((20000 + RANDOM % 20000))
  env "$@" CRANK_AMD_DIST_BACKEND=nccl CRANK_AMD_FORCE_DIST=1 NCCL_DEBUG=WARN timeout 300 \
    python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port $port \
    tests/dp_gpu_worker.py "$out/run$i.npz" lsgan 4 120 bf16 graph 0 7 > "$out/run$i.stdout" 2> "$out/run$i.launcher"
  rc=$?
  last=$(tail -n 1 "$out/run$i.rank0.log" 2>/dev/null)
  echo "run $i rc=$rc last_log_line=[$last]"
  if [ $rc -eq 0 ]; then ok=$((ok+1)); rm -f "$out/run$i.npz.rank0.npz"; else bad=$((bad+1)); fi
done
echo "SUMMARY $* : ok=$ok bad=$bad of $n"

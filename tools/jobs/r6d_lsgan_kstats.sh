#!/bin/bash
# the lsgan step's kernel statistics on the final tree of the fourth session
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; O=$PWD/gpurun_out; mkdir -p $O
rm -rf /tmp/lsp; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/lsp -- python bench.py --trainer lsgan --steps 60 --warmup 10 --no-cpu-baseline --no-extras --no-roofline > /tmp/lsp.log 2>&1 || tail -3 /tmp/lsp.log
python tools/kstats.py /tmp/lsp > $O/round6_i_lsgan_kstats.txt; head -12 $O/round6_i_lsgan_kstats.txt

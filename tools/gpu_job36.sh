#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python tools/vq_phase_cycles.py 2>&1 | tail -1
sed 's/libcrank_hip_vqprof.so/libcrank_hip_vqprof2.so/' tools/vq_phase_cycles.py > /tmp/vqp2.py; timeout 300 python /tmp/vqp2.py 2>&1 | tail -1

#!/bin/bash
# the round's evidence on the tree of the third session
cd "${GRAFT_REPO_ROOT:-.}"
bash tools/round_evidence.sh r6c_ev

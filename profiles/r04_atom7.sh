#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash $R/profiles/r04_sweep.sh "--pot eam --method thread_atom --steps 20 --warmup 5" COMD_EAM_ATOM_LDS_PAD 0 20000

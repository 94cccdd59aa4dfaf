#!/bin/bash
# what the staging prologue alone costs (COMD_EAM_ABLATE=4: the kernels return behind it), both EAM brick kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash $R/profiles/r04_sweep.sh "--pot eam --method thread_atom --steps 20 --warmup 5" COMD_EAM_ABLATE 0 4
bash $R/profiles/r04_sweep.sh "--pot eam --method cta_cell --steps 20 --warmup 5" COMD_EAM_ABLATE 0 4

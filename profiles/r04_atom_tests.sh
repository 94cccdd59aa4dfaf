#!/bin/bash
# the parity legs that touch EAM thread_atom (eam_atom_brick_kernels.h), then its A/B timings
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
python3 -m pytest $R/tests -m gpu -x -q -k "full_size or thread_atom_on_the_brick or hand_over or (thread_atom and (eam or any_cell or recorded or sweep))" > $R/gpurun_out/r04_atom_tests.log 2>&1 || { tail -40 $R/gpurun_out/r04_atom_tests.log; exit 1; }
tail -2 $R/gpurun_out/r04_atom_tests.log
bash $R/profiles/r04_sweep.sh "--pot eam --method thread_atom --steps 20 --warmup 5" COMD_EAM_ATOM_HANDOVER 1 0
bash $R/profiles/r04_sweep.sh "--pot eam --method thread_atom --steps 20 --warmup 5" COMD_EAM_THREAD_ATOM brick cell

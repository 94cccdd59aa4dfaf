#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
bash $R/profiles/r04_sweep.sh "--pot eam --method thread_atom --steps 20 --warmup 5" COMD_EAM_ATOM_HANDOVER 1 0
python3 -m pytest $R/tests -m gpu -x -q -k "thread_atom_on_the_brick or hand_over or overlap or reproducible or hilbert or sweep or (thread_atom and (eam or any_cell or recorded))" > $R/gpurun_out/r04_atom_tests.log 2>&1 || { tail -40 $R/gpurun_out/r04_atom_tests.log; exit 1; }
tail -2 $R/gpurun_out/r04_atom_tests.log

#!/bin/bash
# EAM thread_atom on the brick image (eam_atom_brick_kernels.h) against round 2's kernel, brick shapes; parity tests of the method first
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
python3 -m pytest $R/tests -m gpu -x -q -k "thread_atom or eam or Eam or EAM" > $R/gpurun_out/r04_atom_tests.log 2>&1 || { tail -30 $R/gpurun_out/r04_atom_tests.log; exit 1; }
tail -3 $R/gpurun_out/r04_atom_tests.log
bash $R/profiles/r04_sweep.sh "--pot eam --method thread_atom --steps 20 --warmup 5" COMD_EAM_THREAD_ATOM cell brick
bash $R/profiles/r04_sweep.sh "--pot eam --method thread_atom --steps 20 --warmup 5" COMD_EAM_ATOM_BRICK 4,4 4,5 4,3 4,2 2,2

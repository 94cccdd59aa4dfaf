#!/bin/bash
# lj_overlap.sh: LJ thread_atom step time with the candidate lists of the next run of cells built beside the force kernel (COMD_LJ_OVERLAP = runs)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for k in 1 2 4 8 16; do
  COMD_LJ_OVERLAP=$k timeout -k 10 200 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-variants --no-target-line > gpurun_out/ljov_$k.log 2>&1 || { tail -5 gpurun_out/ljov_$k.log; exit 1; }
  grep '^{"metric' gpurun_out/ljov_$k.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('runs $k ms/step', round(d['ms_per_step'],4), 'kernel', round(r['kernel_ms_per_step'],4), 'eval', r.get('force_evaluation_ms'), 'E/atom', d['energy_per_atom_eV'])"
done

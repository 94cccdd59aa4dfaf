#!/bin/bash
# eam_persist.sh: EAM cta_cell with k workgroups per CU walking the bricks in a loop (COMD_EAM_PERSIST=k; 0 = one workgroup per brick)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for k in 0 4 8 16; do
  COMD_EAM_PERSIST=$k timeout -k 10 200 python3 bench.py --pot eam --method cta_cell --steps 40 --warmup 5 --no-cpu-baseline --no-variants > gpurun_out/eamp_$k.log 2>&1 || { tail -5 gpurun_out/eamp_$k.log; exit 1; }
  grep '^{"metric' gpurun_out/eamp_$k.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('persist $k ms/step', round(d['ms_per_step'],4), 'force', round(r['kernel_ms_per_step'],4), 'E/atom', d['energy_per_atom_eV'])"
done

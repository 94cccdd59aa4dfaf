#!/bin/bash
# soak.sh: long runs of the shipped executable at the BASELINE sizes (energy conservation, no overflow / lost-atom stop): LJ and EAM 80^3 for 2000 steps, LJ 256^3 for 100
R=${GRAFT_REPO_ROOT:-/root/repo}
C=$R/comd-cuda-async_amd/csrc
O=$R/gpurun_out/soak; mkdir -p $O; cd $R
timeout -k 10 300 $C/comd-hip --deviceTimers -x 80 -y 80 -z 80 -N 2000 -n 200 -m thread_atom > $O/lj80_2000_steps.txt 2>&1 || { tail -n 5 $O/lj80_2000_steps.txt; exit 1; }
timeout -k 10 300 $C/comd-hip --deviceTimers -e -x 80 -y 80 -z 80 -N 2000 -n 200 -m cta_cell > $O/eam80_2000_steps.txt 2>&1 || { tail -n 5 $O/eam80_2000_steps.txt; exit 1; }
timeout -k 10 500 $C/comd-hip --deviceTimers -x 256 -y 256 -z 256 -N 100 -n 10 -m thread_atom > $O/lj256_100_steps.txt 2>&1 || { tail -n 5 $O/lj256_100_steps.txt; exit 1; }
rm -f CoMD-hip*.yaml
for f in lj80_2000_steps eam80_2000_steps lj256_100_steps; do echo "== $f"; grep -E "^ +[0-9]+ +[0-9.]+ +-" $O/$f.txt | sed -n '1p;$p'; grep -i "atom update rate\|atomUpdatesPerSec\|Final energy\|eFinal\|Max Link" $O/$f.txt | head -5; done

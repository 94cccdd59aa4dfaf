#!/bin/bash
# odd configurations of the EAM list method against cta_cell on the same input: final total energy per atom must agree (both are held to the oracle by the tests)
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/comd-cuda-async_amd/csrc/comd-hip
run() { $C -d $R/pots -e "$@" 2>&1 | awk '/^ *[0-9]+ +[0-9.]+ +-?[0-9.]+/ {last=$0} /eFinal\/eInitial/ {r=$0} END {print last; print r}'; }
for extra in "-x 12 -y 12 -z 12 -T 3000 -N 200 -n 50" "-x 6 -y 6 -z 6 -N 100 -n 50" "-x 20 -y 14 -z 11 -r 0.2 -N 100 -n 50 -H" "-x 16 -y 16 -z 16 -N 100 -n 50 -a 1" "-x 12 -y 12 -z 12 -N 100 -n 50 --maxAtoms 128" "-x 12 -y 12 -z 12 -N 100 -n 50 -t setfl -p Cu01.eam.alloy" "-x 12 -y 12 -z 12 -N 100 -n 50 -P"; do
  echo "== $extra"
  echo "  nl : $(run $extra -m thread_atom_nl | head -1)"
  echo "  cta: $(run $extra -m cta_cell | head -1)"
done
rm -f CoMD-hip*.yaml

#!/bin/bash
# usage: bash profiles/r04_sweep.sh "bench args" ENVNAME v1 v2 ...   -- one bench.py run per value of the environment variable, one line each
R=${GRAFT_REPO_ROOT:-$(pwd)}
ARGS=$1; NAME=$2; shift 2
for v in "$@"; do
  env $NAME="$v" python3 $R/bench.py $ARGS --no-cpu-baseline --no-variants --no-target-line > /tmp/sweep.json 2> /tmp/sweep.err || { echo "$NAME=$v FAILED"; tail -3 /tmp/sweep.err; continue; }
  python3 -c "import json; d=json.loads(open('/tmp/sweep.json').read().strip().splitlines()[-1]); print('$NAME=$v', 'ms/step %.4f' % d['ms_per_step'], 'force %.4f' % d['roofline']['kernel_ms_per_step'], 'E %.9f' % d['energy_per_atom_eV'])"
done

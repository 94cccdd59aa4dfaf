#!/bin/bash
# one-GPU rehearsal of what bench.py --gpus N runs: every halo message through a one-rank RCCL communicator, overlap mode, with the sized-vs-handshake self-check
R=${GRAFT_REPO_ROOT:-$(pwd)}
for a in "--pot lj" "--pot eam" "--pot eam --method thread_atom_nl" "--pot lj --method thread_atom_nl"; do
  COMD_LOOPBACK_TRANSPORT=1 python3 $R/bench.py $a --async-halo 1 --no-variants --no-cpu-baseline --no-target-line > /tmp/lb.json 2> /tmp/lb.err; rc=$?
  python3 -c "import json; d=json.loads(open('/tmp/lb.json').read().strip().splitlines()[-1]); print('$a rc=$rc', 'ms/step %.3f' % d['ms_per_step'], 'sized_matches_handshake', d.get('sized_matches_handshake'), 'handshake ms %.3f' % d['handshake_run']['ms_per_step'], 'eF/eI %.9f' % d['eFinal_over_eInitial'])" || tail -5 /tmp/lb.err
done

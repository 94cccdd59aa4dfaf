#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { env "$@" python3 $R/bench.py $ARGS --no-variants --no-cpu-baseline --no-target-line > /tmp/lb.json 2> /tmp/lb.err; python3 -c "import json; d=json.loads(open('/tmp/lb.json').read().strip().splitlines()[-1]); print('$ARGS | $*', '| ms/step %.3f force %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms_per_step']))" || tail -5 /tmp/lb.err; }
for ARGS in "--pot eam" "--pot eam --async-halo 1"; do
  run COMD_X=0
  run COMD_LOOPBACK_TRANSPORT=1
  run COMD_HALO_MIRROR=0
done

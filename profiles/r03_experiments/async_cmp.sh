#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for pot in lj eam; do for a in 0 1; do for lb in 0 1; do
  COMD_LOOPBACK_TRANSPORT=$lb python3 bench.py --pot $pot --async-halo $a --steps 40 --warmup 5 --no-variants --no-cpu-baseline --no-target-line 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$pot async=$a loopback=$lb ms/step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms_per_step'],4), 'eval', round(d['roofline'].get('force_evaluation_ms',0),4))"
done; done; done

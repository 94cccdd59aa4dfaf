#!/bin/bash
# usage (on the GPU box, from the repo root): bash profiles/r04_prof.sh TAG "bench args" [ENV=VALUE ...]
# rocprofv3 --kernel-trace --stats of one bench.py run; the stats CSV lands in gpurun_out/r04p_TAG_kernel_stats.csv, the bench line in gpurun_out/r04p_TAG.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; ARGS=$2; shift 2
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/r04p_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r04p_$TAG -o x -- python3 $R/bench.py $ARGS --no-cpu-baseline --no-variants --no-target-line > $R/gpurun_out/r04p_$TAG.json 2> $R/gpurun_out/r04p_$TAG.err
rc=$?
f=$(find /tmp/r04p_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $R/gpurun_out/r04p_${TAG}_kernel_stats.csv
echo "$TAG rc=$rc $(python3 -c "import json,sys; d=json.loads(open('$R/gpurun_out/r04p_$TAG.json').read().strip().splitlines()[-1]); print('ms/step', d['ms_per_step'], 'force', d['roofline'].get('kernel_ms_per_step'))" 2>&1)"
[ -n "$f" ] && python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print("   %-60s calls %5s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1000))
P
exit $rc

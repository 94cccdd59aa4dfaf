#!/bin/bash
# stats_run.sh TAG "ENV=.. ENV2=.." "bench args"
# One rocprofv3 --kernel-trace --stats pass of bench.py on the MI355X box; prints the top kernels (name, calls, avg us).
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; ENVS=$2; ARGS=$3
O=$R/gpurun_out/stats_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for e in $ENVS; do export $e; done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o out -- python3 $R/bench.py $ARGS --no-cpu-baseline --no-variants > $O/run.log 2>&1 || { echo "run failed"; tail -5 $O/run.log; exit 1; }
find $O -name "*kernel_trace.csv" -delete
python3 - "$O" "$TAG" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/out_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("==", sys.argv[2])
for r in rows[:5]:
    print("  %-60s calls %5s avg %10.1f us  %5.1f%%" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
grep '^{"metric' $O/run.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ms/step', round(d['ms_per_step'],4), 'force ms/step', round(d['roofline']['kernel_ms_per_step'],4), 'E/atom', d['energy_per_atom_eV'])"

#!/bin/bash
# pmc_run.sh TAG "ENV=.. ENV2=.." "bench args" "CTR1 CTR2 ..." ["CTR9 ..." more passes]
# Runs on the MI355X box (gpurun): one rocprofv3 --pmc pass per counter group (--kernel-trace only, as the pool requires),
# then prints the per-kernel average of every counter.  Output under gpurun_out/pmc_TAG/.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; ENVS=$2; ARGS=$3; shift 3
O=$R/gpurun_out/pmc_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for e in $ENVS; do export $e; done
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/p$i -o out -- python3 $R/bench.py $ARGS --no-cpu-baseline --no-variants > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/p$i.log; }
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections, json
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.Counter())
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
out = {k: {c: acc[k][c] / cnt[k][c] for c in acc[k]} | {"launches": max(cnt[k].values())} for k in acc}
json.dump(out, open(O + "/summary.json", "w"), indent=1)
for k in sorted(out, key=lambda k: -out[k].get("SQ_WAVE_CYCLES", out[k].get("SQ_BUSY_CYCLES", 0)))[:6]:
    print(k, json.dumps({c: round(v, 1) for c, v in out[k].items()}))
PY

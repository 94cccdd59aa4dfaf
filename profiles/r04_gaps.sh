#!/bin/bash
# usage: bash profiles/r04_gaps.sh TAG "bench args" [ENV=VALUE ...] -- kernel trace of one bench run; prints, for the steady state, the time the device spent in kernels and idle between them
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; ARGS=$2; shift 2
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/r04g_$TAG
rocprofv3 --kernel-trace --output-format csv -d /tmp/r04g_$TAG -o x -- python3 $R/bench.py $ARGS --no-cpu-baseline --no-variants --no-target-line > /tmp/r04g_$TAG.json 2> /tmp/r04g_$TAG.err
f=$(find /tmp/r04g_$TAG -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$TAG" <<'P'
import csv, sys, collections
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in csv.DictReader(open(sys.argv[1]))), key=lambda r: r[0])
# steady state: the last 40 % of the trace
lo = int(len(rows) * 0.6)
rows = rows[lo:]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = collections.Counter(); gapn = collections.Counter()
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    g = max(0, s1 - e0)
    gaps[(n0[:38], n1[:38])] += g; gapn[(n0[:38], n1[:38])] += 1
print(sys.argv[2], "kernels %d  span %.2f ms  busy %.2f ms  idle %.1f %%" % (len(rows), span / 1e6, busy / 1e6, 100.0 * (span - busy) / span))
for k, v in gaps.most_common(8):
    print("   idle %7.1f us total, %6.1f us each x %d   after %-38s before %s" % (v / 1e3, v / 1e3 / gapn[k], gapn[k], k[0], k[1]))
P

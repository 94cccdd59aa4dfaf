#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04atom
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o out -- python3 $R/bench.py --pot eam --method thread_atom --no-cpu-baseline --no-variants --no-target-line --steps 50 --warmup 5 > $O/stats.log 2>&1 &&
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/sq1 -o out -- python3 $R/bench.py --pot eam --method thread_atom --no-cpu-baseline --no-variants --no-target-line --steps 6 --warmup 2 > $O/sq1.log 2>&1 &&
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $O/sq2 -o out -- python3 $R/bench.py --pot eam --method thread_atom --no-cpu-baseline --no-variants --no-target-line --steps 6 --warmup 2 > $O/sq2.log 2>&1
find $O -name "*kernel_stats.csv" | head -1 | xargs head -8
python3 - <<'PY'
import csv,glob,os,collections
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r04atom"
for d in ("sq1","sq2"):
    for f in glob.glob(O+"/"+d+"/**/*counter_collection.csv", recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:60]
            if "atom_brick" not in k: continue
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
        for k,v in acc.items():
            print(k, {c: "%.3g"%x for c,x in v.items()})
PY

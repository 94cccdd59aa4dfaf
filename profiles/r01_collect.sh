set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -2 $O/smoke.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "default bench done"
./comd-cuda-async_amd/csrc/comd-hip -x 20 -y 20 -z 20 -m thread_atom_nl > $O/comd_hip_lj20_nl_stdout.txt 2>&1
./comd-cuda-async_amd/csrc/comd-hip -e -x 20 -y 20 -z 20 -m thread_atom_nl > $O/comd_hip_eam20_nl_stdout.txt 2>&1
./comd-cuda-async_amd/csrc/comd-hip -e -t setfl -x 20 -y 20 -z 20 -m cta_cell > $O/comd_hip_setfl20_stdout.txt 2>&1
./comd-cuda-async_amd/csrc/comd-hip -x 20 -y 20 -z 20 -m cta_cell -L > $O/comd_hip_lj20_pairlist_stdout.txt 2>&1
rm -f CoMD-hip*.yaml
cd /tmp && export TMPDIR=/tmp
for c in "lj thread_atom" "lj thread_atom_nl" "eam cta_cell" "eam thread_atom_nl"; do
  set -- $c
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$1_$2 -o out -- python3 $R/bench.py --pot $1 --method $2 --no-cpu-baseline --no-variants --steps 100 --warmup 10 > $O/stats_$1_$2.log 2>&1
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$1_$2_$ctr -o out -- python3 $R/bench.py --pot $1 --method $2 --no-cpu-baseline --no-variants --steps 10 --warmup 3 > $O/pmc_$1_$2_$ctr.log 2>&1
  done
  echo "profiled $1 $2"
done
python3 - <<'PY'
import csv,glob,os,collections,json
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/final"
out={}
for d in sorted(glob.glob(O+"/pmc_*")):
    if not os.path.isdir(d): continue
    f=glob.glob(d+"/**/*counter_collection.csv", recursive=True)
    if not f: continue
    acc=collections.defaultdict(float); cnt=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"].split("(")[0]
        if "Force" in k or "Build" in k:
            acc[k]+=float(r["Counter_Value"]); cnt[k]+=1
    out[os.path.basename(d)]={k:{"sum":v,"launches":cnt[k],"per_launch":v/cnt[k]} for k,v in acc.items()}
json.dump(out, open(O+"/pmc_summary.json","w"), indent=1)
PY
echo all done

# Round-4 evidence, collected in TWO gpurun calls on the MI355X box:
#   gpurun --timeout 1190 -- 'bash profiles/r04_collect.sh 1'     tests, smoke, benches, comd-hip stdouts (host and device timers), loopback legs
#   gpurun --timeout 1190 -- 'bash profiles/r04_collect.sh 2'     rocprofv3 stats + PMC passes
# then, here:  python3 profiles/r04_summarize.py   (copies what is to be tracked into profiles/ and writes r04_summary.md)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04final
PHASE=${1:-1}
mkdir -p $O
cd $R
if [ "$PHASE" = "1" ]; then
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
fi
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
T0=$SECONDS; python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench: $((SECONDS - T0)) s wall" | tee $O/bench_default.time
echo "default bench done"
C=./comd-cuda-async_amd/csrc
$C/comd-hip -x 20 -y 20 -z 20 -m thread_atom > $O/comd_hip_lj20_stdout.txt 2>&1
$C/comd-hip --deviceTimers -x 20 -y 20 -z 20 -m thread_atom > $O/comd_hip_lj20_device_timers_stdout.txt 2>&1
$C/comd-hip --deviceTimers -x 40 -y 40 -z 40 -m thread_atom > $O/comd_hip_lj40_device_timers_stdout.txt 2>&1
$C/comd-hip -e -x 20 -y 20 -z 20 -m cta_cell > $O/comd_hip_eam20_stdout.txt 2>&1
$C/comd-hip --deviceTimers -e -x 40 -y 40 -z 40 -m cta_cell > $O/comd_hip_eam40_device_timers_stdout.txt 2>&1
$C/comd-hip -e -x 20 -y 20 -z 20 -m thread_atom > $O/comd_hip_eam20_thread_atom_stdout.txt 2>&1
$C/comd-hip -x 20 -y 20 -z 20 -m thread_atom_nl > $O/comd_hip_lj20_nl_stdout.txt 2>&1
$C/comd-hip -e -x 20 -y 20 -z 20 -m thread_atom_nl > $O/comd_hip_eam20_nl_stdout.txt 2>&1
$C/comd-hip -e -t setfl -p Cu01.eam.alloy -x 20 -y 20 -z 20 -m cta_cell > $O/comd_hip_setfl20_stdout.txt 2>&1
$C/comd-hip -x 20 -y 20 -z 20 -m cta_cell -L > $O/comd_hip_lj20_pairlist_stdout.txt 2>&1
$C/comd-hip-sp -x 20 -y 20 -z 20 -m thread_atom > $O/comd_hip_sp_lj20_stdout.txt 2>&1
$C/comd-hip-sp -e -x 20 -y 20 -z 20 -m cta_cell > $O/comd_hip_sp_eam20_stdout.txt 2>&1
rm -f CoMD-hip*.yaml
echo "comd-hip runs done"
COMD_LOOPBACK_TRANSPORT=1 python bench.py --no-variants --no-cpu-baseline > $O/loopback_lj.json 2>/dev/null
COMD_LOOPBACK_TRANSPORT=1 COMD_HALO_HANDSHAKE=1 python bench.py --no-target-line --no-variants --no-cpu-baseline > $O/loopback_lj_handshake.json 2>/dev/null
COMD_LOOPBACK_TRANSPORT=1 python bench.py --pot eam --no-variants --no-cpu-baseline > $O/loopback_eam.json 2>/dev/null
COMD_LOOPBACK_TRANSPORT=1 COMD_HALO_HANDSHAKE=1 python bench.py --pot eam --no-variants --no-cpu-baseline > $O/loopback_eam_handshake.json 2>/dev/null
COMD_LOOPBACK_TRANSPORT=1 python bench.py --async-halo 1 --no-variants --no-cpu-baseline --no-target-line > $O/loopback_lj_overlap.json 2>/dev/null
COMD_LOOPBACK_TRANSPORT=1 python bench.py --pot eam --async-halo 1 --no-variants --no-cpu-baseline > $O/loopback_eam_overlap.json 2>/dev/null
python bench.py --no-variants --no-cpu-baseline --no-target-line > $O/plain_lj.json 2>/dev/null
python bench.py --pot eam --no-variants --no-cpu-baseline > $O/plain_eam.json 2>/dev/null
COMD_HALO_MIRROR=0 python bench.py --no-variants --no-cpu-baseline --no-target-line > $O/plain_lj_messages.json 2>/dev/null
COMD_HALO_MIRROR=0 python bench.py --pot eam --no-variants --no-cpu-baseline > $O/plain_eam_messages.json 2>/dev/null
COMD_HALO_MIRROR=0 python bench.py --pot eam --method thread_atom_nl --no-variants --no-cpu-baseline > $O/plain_eam_nl_messages.json 2>/dev/null
python bench.py --pot eam --method thread_atom_nl --no-variants --no-cpu-baseline > $O/plain_eam_nl.json 2>/dev/null
COMD_EAM_NL=lds python bench.py --pot eam --method thread_atom_nl --no-variants --no-cpu-baseline > $O/plain_eam_nl_round3_kernel.json 2>/dev/null
python bench.py --pot eam --method thread_atom --no-variants --no-cpu-baseline > $O/plain_eam_thread_atom.json 2>/dev/null
COMD_EAM_THREAD_ATOM=cell python bench.py --pot eam --method thread_atom --no-variants --no-cpu-baseline > $O/plain_eam_thread_atom_round2_kernel.json 2>/dev/null
COMD_EAM_ATOM_HANDOVER=0 python bench.py --pot eam --method thread_atom --no-variants --no-cpu-baseline > $O/plain_eam_thread_atom_no_handover.json 2>/dev/null
COMD_LOOPBACK_TRANSPORT=1 python bench.py --pot eam --method thread_atom --async-halo 1 --no-variants --no-cpu-baseline > $O/loopback_eam_thread_atom_overlap.json 2>/dev/null
COMD_LOOPBACK_TRANSPORT=1 python bench.py --pot eam --method thread_atom_nl --async-halo 1 --no-variants --no-cpu-baseline > $O/loopback_eam_nl_overlap.json 2>/dev/null
echo "loopback benches done"
exit 0
fi
cd /tmp && export TMPDIR=/tmp
for c in "lj thread_atom" "lj thread_atom_nl" "lj cta_cell" "eam cta_cell" "eam thread_atom_nl" "eam thread_atom"; do
  set -- $c
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$1_$2 -o out -- python3 $R/bench.py --pot $1 --method $2 --no-cpu-baseline --no-variants --no-target-line --steps 100 --warmup 10 > $O/stats_$1_$2.log 2>&1
  echo "stats $1 $2"
done
COMD_LOOPBACK_TRANSPORT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_loopback_lj -o out -- python3 $R/bench.py --no-cpu-baseline --no-variants --steps 50 --warmup 5 > $O/stats_loopback_lj.log 2>&1
for c in "lj thread_atom" "lj thread_atom_nl" "eam cta_cell" "eam thread_atom_nl" "eam thread_atom"; do
  set -- $c
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$1_$2_$ctr -o out -- python3 $R/bench.py --pot $1 --method $2 --no-cpu-baseline --no-variants --no-target-line --steps 10 --warmup 3 > $O/pmc_$1_$2_$ctr.log 2>&1
  done
  echo "pmc $1 $2"
done
for c in "lj thread_atom" "eam cta_cell" "eam thread_atom_nl" "lj thread_atom_nl" "eam thread_atom"; do
  set -- $c
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_$1_$2_SQ1 -o out -- python3 $R/bench.py --pot $1 --method $2 --no-cpu-baseline --no-variants --no-target-line --steps 6 --warmup 2 > $O/pmc_$1_$2_SQ1.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $O/pmc_$1_$2_SQ2 -o out -- python3 $R/bench.py --pot $1 --method $2 --no-cpu-baseline --no-variants --no-target-line --steps 6 --warmup 2 > $O/pmc_$1_$2_SQ2.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_$1_$2_TCC -o out -- python3 $R/bench.py --pot $1 --method $2 --no-cpu-baseline --no-variants --no-target-line --steps 6 --warmup 2 > $O/pmc_$1_$2_TCC.log 2>&1
  echo "sq pmc $1 $2"
done
python3 $R/profiles/r04_summarize.py --hashes > $O/kernel_source_hashes.json
python3 - <<'PY'
import csv,glob,os,collections,json
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r04final"
out={}
for d in sorted(glob.glob(O+"/pmc_*")):
    if not os.path.isdir(d): continue
    f=glob.glob(d+"/**/*counter_collection.csv", recursive=True)
    if not f: continue
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")
        if "Force" in k or "Build" in k or "Candidates" in k or "Pack" in k or "MarkCells" in k:
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k][r["Counter_Name"]]+=1
    out[os.path.basename(d)]={k:{c:{"per_launch":acc[k][c]/cnt[k][c],"launches":cnt[k][c]} for c in acc[k]} for k in acc}
    # raw CSVs are large: keep the summary only
    for g in glob.glob(d+"/**/*counter_collection.csv", recursive=True)+glob.glob(d+"/**/*kernel_trace.csv", recursive=True): os.remove(g)
json.dump(out, open(O+"/pmc_summary.json","w"), indent=1)
for d in glob.glob(O+"/stats_*"):
    for g in glob.glob(d+"/**/*kernel_trace.csv", recursive=True): os.remove(g)
PY
echo all done

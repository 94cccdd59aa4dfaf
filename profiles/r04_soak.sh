#!/bin/bash
# 2000 steps of EAM 80^3 thread_atom on the brick image, then cta_cell: energy drift (eFinal_over_eInitial), bricks in the fall-back, step time over a long window
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
python3 $R/bench.py --pot eam --method thread_atom --steps 2000 --warmup 10 --no-cpu-baseline --no-variants --no-target-line > $R/gpurun_out/r04_soak_eam80_thread_atom_2000_steps.json 2> $R/gpurun_out/r04_soak_thread_atom.err || { tail -5 $R/gpurun_out/r04_soak_thread_atom.err; exit 1; }
python3 $R/bench.py --pot eam --method cta_cell --steps 2000 --warmup 10 --no-cpu-baseline --no-variants --no-target-line > $R/gpurun_out/r04_soak_eam80_cta_cell_2000_steps.json 2> $R/gpurun_out/r04_soak_cta.err || { tail -5 $R/gpurun_out/r04_soak_cta.err; exit 1; }
python3 - <<PY
import json
for m in ("thread_atom", "cta_cell"):
    d = json.loads(open("$R/gpurun_out/r04_soak_eam80_%s_2000_steps.json" % m).read().strip().splitlines()[-1])
    print(m, "ms/step %.4f" % d["ms_per_step"], "force %.4f" % d["roofline"]["kernel_ms_per_step"], "eF/eI %.9f" % d["eFinal_over_eInitial"], d["force_path"])
PY

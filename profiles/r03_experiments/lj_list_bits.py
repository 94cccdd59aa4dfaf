import os, sys, importlib, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
gpu = importlib.import_module("comd-cuda-async_amd")
from test_gpu_parity import _args
out = []
for prune in ("1", "0"):
    os.environ["COMD_LJ_PRUNE"] = prune
    with gpu.Simulation(_args((70, 7, 7), 0, 0.2, "thread_atom")) as sim:
        sim.step(2)
        out.append((sim.gather(2).copy(), sim.gather(3).copy()))
d = np.abs(out[0][0] - out[1][0])
print("atoms", len(d), "differing", int((d.max(1) > 0).sum()), "max abs diff", d.max(), "max |f|", np.abs(out[1][0]).max())
print("energy differing", int((out[0][1] != out[1][1]).sum()), np.abs(out[0][1] - out[1][1]).max())

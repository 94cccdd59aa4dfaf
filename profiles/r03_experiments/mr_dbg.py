import sys
sys.path.insert(0, 'tests')
import test_multirank as t
for grid, meth, a, env in (((2,1,2),"cta_cell+H",1,{"COMD_EAM_BRICK":"2,2"}), ((2,1,2),"cta_cell",1,{"COMD_EAM_BRICK":"2,2"}), ((2,1,2),"cta_cell",0,{"COMD_EAM_BRICK":"2,2"}), ((1,1,1),"cta_cell",0,{"COMD_EAM_BRICK":"2,2"})):
    try:
        outs = t._launch("gpu", grid, 1, 12, extra=(meth, a), env_extra=env)
        print(grid, meth, a, env, "OK" if "gpu-mode OK" in outs[0] else "??")
    except AssertionError as e:
        print(grid, meth, a, env, "FAIL", str(e)[-1800:])

import os, sys
sys.path.insert(0, '/root/repo')
if len(sys.argv) > 1 and sys.argv[1] == 'stamps':
    os.environ["RELP_AMD_LIB"] = "/root/repo/relp_amd/librelp_amd_stamps.so"
import relp_amd
for name in ["GREENBEA", "BNL2", "CYCLE", "GREENBEB", "25FV47"]:
    for rep in range(3):
        try:
            s = relp_amd.Solver(carry=int(os.environ.get("RELP_FLAKY_CARRY", "1"))).load_mps('/root/repo/data/netlib/%s.SIF' % name)
            r = s.solve_relaxation()
            print(name, rep, r.kind, r.objective, r.pivots_phase_one + r.pivots_phase_two, r.refactors, flush=True)
            s.close()
        except Exception as e:
            print(name, rep, "FAILED", e, flush=True)

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relp_amd
from relp_amd.workloads import max_flow_graph
V, E = int(sys.argv[1]), int(sys.argv[2])
tail, head, cap = max_flow_graph(V, E)
model = relp_amd.Model.max_flow(V, list(zip(tail.tolist(), head.tolist(), cap.tolist())), 0, V - 1)
s = relp_amd.Solver(implicit_bounds=1).load_model(model)
s.begin_phase_one()
print("iterate:", s.iterate(200))
print("iterate:", s.iterate(200))
try:
    print(s.profile_kernel(0, 20))
except Exception as e:
    print("profile failed:", e)
r = s.solve_relaxation()
print("solve:", r.kind, r.pivots_phase_one, r.pivots_phase_two, r.objective)
s.begin_phase_one()
print("iterate after solve:", s.iterate(200))
try:
    print(s.profile_kernel(0, 50), s.profile_kernel(0, 200), s.profile_kernel(1, 200), s.profile_kernel(2, 200))
except Exception as e:
    print("profile failed:", e)

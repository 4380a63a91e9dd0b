"""One-off fuzzing on the GPU box: random general forms (free / bounded / shifted variables, every row kind, maximisation)
solved WITH the presolve on the device path against the exact oracle WITHOUT presolve: the optimum must be the same rational,
and whatever the presolve refuses (infeasible / unbounded / solved completely) must match the oracle's verdict.

    python tests/fuzz/fuzz_presolve_gpu.py [first_seed] [count]
"""
import os, random, sys, time
from fractions import Fraction
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import relp_amd
from relp_oracle import FiniteOptimum, Infeasible, Unbounded, solve_relaxation
from test_host_general_form import oracle_standard_form, random_general_form

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
bad, refused, solved = [], 0, 0
start = time.time()
for seed in range(first, first + count):
    rng = random.Random(660000 + seed)
    form = random_general_form(rng, rng.randint(2, 8), rng.randint(2, 9))
    columns, kinds, b, variables, maximize, fixed_cost = form
    general, data = oracle_standard_form(*form)
    try:
        expected = solve_relaxation(data)
    except AssertionError:
        continue
    if isinstance(expected, FiniteOptimum):
        want = general.objective_of(data.reconstruct_solution(expected.solution))
    if maximize:
        # The reference negates the costs of a maximisation (general_form/mod.rs:623-633) but not the fixed cost it has
        # accumulated from the shifts with the original signs (:535), so the value it reports for a maximisation depends on
        # the shifts -- and the presolve changes them.  For those the yardstick is the oracle WITH its presolve.
        from relp_oracle.presolve import Infeasible as PI, Unbounded as PU
        try:
            general, data = oracle_standard_form(*form, presolve=True)
            if not general.variables or not general.b:
                continue
            expected = solve_relaxation(data)
            if isinstance(expected, FiniteOptimum):
                want = general.objective_of(data.reconstruct_solution(expected.solution))
        except (PI, PU, AssertionError, IndexError):
            continue
    try:
        model = relp_amd.Model.from_general_form(columns, kinds, b, variables, maximize=maximize, fixed_cost=fixed_cost, presolve=True)
    except relp_amd.RelpError as error:
        refused += 1
        text = str(error)
        # "solved completely" is fine for a finite optimum; infeasible / unbounded verdicts must agree with the oracle
        if "infeasible" in text.lower() and not isinstance(expected, Infeasible):
            bad.append((seed, "presolve says infeasible", type(expected).__name__))
        if "unbounded" in text.lower() and not isinstance(expected, (Unbounded, Infeasible)):
            bad.append((seed, "presolve says unbounded", type(expected).__name__))
        continue
    solver = relp_amd.Solver(certify=1).load_model(model)
    result = solver.solve_relaxation()
    solved += 1
    if isinstance(expected, Infeasible):
        ok = result.kind == relp_amd.INFEASIBLE
    elif isinstance(expected, Unbounded):
        ok = result.kind in (relp_amd.UNBOUNDED, relp_amd.INFEASIBLE) and result.kind == relp_amd.UNBOUNDED
    else:
        ok = result.kind == relp_amd.FINITE_OPTIMUM and result.certified == 1 and Fraction(solver.objective_exact()) == want
    if not ok:
        bad.append((seed, "kind %d certified %d" % (result.kind, result.certified), type(expected).__name__))
        print("MISMATCH", bad[-1], flush=True)
    solver.close()
print("checked %d seeds in %.1f s (%d solved on the device, %d refused by the presolve): %d mismatches %s" % (
    count, time.time() - start, solved, refused, len(bad), bad[:20]))

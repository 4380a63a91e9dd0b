"""``relp_model_from_general_form`` (GeneralForm::new + [presolve +] standardize + derive_matrix_data) against the oracle.

The oracle's ``GeneralForm`` is pinned here by the two known-answer tests the reference holds for these transformations
(general_form/mod.rs:1131-1328: ``shift_variables``, ``make_b_non_negative``); the C++ host is then compared with the oracle
on random general forms with every variable and row kind.  CPU only; the ``gpu`` test solves a few of them."""
import random
from fractions import Fraction as F

import pytest

import relp_amd
from relp_oracle.mps import GeneralForm
from relp_oracle.provider import Variable


def test_oracle_reproduces_shift_variables():
    """general_form/mod.rs:1131-1227."""
    general = GeneralForm("Minimize", [[(0, F(1)), (1, F(2))], [(1, F(1))]], ["Greater", "Less"], [F(2), F(8)],
                          [Variable(1, lower_bound=None), Variable(3, lower_bound=F(5, 2))], ["XONE", "XTWO"])
    general.fixed_cost = F(1)
    general._transform_variables()
    assert general.fixed_cost == F(1) + F(3) * F(5, 2)
    assert general.columns == [[(0, F(1)), (1, F(2))], [(1, F(1))], [(0, F(-1)), (1, F(-2))]]
    assert general.constraint_types == ["Greater", "Less"]
    assert general.b == [F(2), F(11, 2)]
    assert [(v.cost, v.lower_bound, v.upper_bound, v.shift, v.flipped) for v in general.variables] == [
        (F(1), F(0), None, F(0), False), (F(3), F(0), None, F(-5, 2), False), (F(-1), F(0), None, F(0), False)]
    assert general.free_pairs == {0: 2}


def test_oracle_reproduces_make_b_non_negative():
    """general_form/mod.rs:1229-1285."""
    general = GeneralForm("Minimize", [[(0, F(2))]], ["Equal"], [F(-1)], [Variable(1, lower_bound=None)], ["XONE"])
    general._make_b_non_negative()
    assert general.columns == [[(0, F(-2))]] and general.b == [F(1)] and general.constraint_types == ["Equal"]


def random_general_form(rng, m, n):
    dense = [[rng.choice([0, 0, 0, 1, 2, -1, -3, F(1, 2), F(-2, 3)]) for _ in range(n)] for _ in range(m)]
    for i in range(m):  # no empty rows (the reference's presolve would remove them; without presolve they are legal but dull)
        if not any(dense[i]):
            dense[i][rng.randrange(n)] = 1
    columns = [[(i, F(dense[i][j])) for i in range(m) if dense[i][j] != 0] for j in range(n)]
    kinds = [rng.choice(["Equal", "Less", "Greater", ("Range", F(rng.randint(1, 6), rng.choice([1, 2])))]) for _ in range(m)]
    b = [F(rng.randint(-9, 12), rng.choice([1, 1, 3])) for _ in range(m)]
    variables = []
    for _ in range(n):
        shape = rng.choice(["lower", "lower", "free", "upper", "both", "shifted"])
        cost = F(rng.randint(-5, 5), rng.choice([1, 1, 2]))
        if shape == "lower":
            variables.append((cost, 0, None))
        elif shape == "free":
            variables.append((cost, None, None))
        elif shape == "upper":
            variables.append((cost, None, F(rng.randint(-3, 8))))
        elif shape == "both":
            low = F(rng.randint(-4, 3), rng.choice([1, 2]))
            variables.append((cost, low, low + rng.randint(0, 7)))
        else:
            variables.append((cost, F(rng.randint(-4, 6), 3), None))
    return columns, kinds, b, variables, rng.random() < 0.3, F(rng.randint(-3, 3), rng.choice([1, 4]))


def oracle_standard_form(columns, kinds, b, variables, maximize, fixed_cost, presolve=False):
    general = GeneralForm("Maximize" if maximize else "Minimize", columns, kinds, b,
                          [Variable(c, lower_bound=lo, upper_bound=up) for c, lo, up in variables],
                          ["X%d" % j for j in range(len(variables))])
    general.fixed_cost = F(fixed_cost)
    if presolve:
        from relp_oracle.presolve import presolve as run_presolve
        run_presolve(general)
    counts = general.standardize()
    return general, general.derive_matrix_data(counts)


def assert_same(model, general, data):
    assert model.nr_rows == data.nr_rows() and model.nr_columns == data.nr_columns()
    assert model.group_counts == [data.nr_equality, data.nr_range, data.nr_upper, data.nr_lower]
    assert model.pivot_element_indices() == data.pivot_element_indices()
    for j in range(model.nr_columns):
        assert [(i, F(num, den)) for i, num, den in model.column_exact(j)] == data.column(j), j
        assert model.cost_value(j) == pytest.approx(float(data.cost_value(j)), rel=1e-15, abs=0)
    assert list(model.right_hand_side()) == pytest.approx([float(v) for v in data.right_hand_side()], rel=1e-15, abs=0)
    assert model.fixed_cost() == pytest.approx(float(general.fixed_cost), rel=1e-15, abs=0)


@pytest.mark.parametrize("seed", range(40))
def test_host_equals_oracle_on_random_general_forms(seed):
    rng = random.Random(7000 + seed)
    form = random_general_form(rng, rng.randint(2, 7), rng.randint(2, 8))
    general, data = oracle_standard_form(*form)
    columns, kinds, b, variables, maximize, fixed_cost = form
    model = relp_amd.Model.from_general_form(columns, kinds, b, variables, maximize=maximize, fixed_cost=fixed_cost)
    assert_same(model, general, data)
    assert model.original_variables() == (len(variables), 0)


@pytest.mark.parametrize("seed", range(25))
def test_host_equals_oracle_with_presolve(seed):
    from relp_oracle.presolve import Infeasible as PresolveInfeasible, Unbounded as PresolveUnbounded
    rng = random.Random(8000 + seed)
    form = random_general_form(rng, rng.randint(3, 7), rng.randint(3, 8))
    columns, kinds, b, variables, maximize, fixed_cost = form
    try:
        general, data = oracle_standard_form(*form, presolve=True)
    except (PresolveInfeasible, PresolveUnbounded):
        with pytest.raises(relp_amd.RelpError):  # infeasible / unbounded / solved completely: both sides must refuse
            relp_amd.Model.from_general_form(columns, kinds, b, variables, maximize=maximize, fixed_cost=fixed_cost, presolve=True)
        return
    if not general.variables or not general.b:
        with pytest.raises(relp_amd.RelpError):
            relp_amd.Model.from_general_form(columns, kinds, b, variables, maximize=maximize, fixed_cost=fixed_cost, presolve=True)
        return
    model = relp_amd.Model.from_general_form(columns, kinds, b, variables, maximize=maximize, fixed_cost=fixed_cost, presolve=True)
    assert_same(model, general, data)
    assert model.original_variables() == (len(variables), len(general.removed))


def test_argument_errors():
    columns, kinds, b, variables = [[(0, 1)], [(0, 1)]], ["Less"], [4], [(1, 0, None), (1, 0, None)]
    relp_amd.Model.from_general_form(columns, kinds, b, variables)
    with pytest.raises(relp_amd.RelpError):
        relp_amd.Model.from_general_form([[(1, 1)], [(0, 1)]], kinds, b, variables)          # row out of range
    with pytest.raises(relp_amd.RelpError):
        relp_amd.Model.from_general_form([[(0, 1), (0, 2)], [(0, 1)]], kinds, b, variables)  # duplicate row in a column
    with pytest.raises(relp_amd.RelpError):
        relp_amd.Model.from_general_form(columns, kinds, b, [(1, 3, 2), (1, 0, None)])       # lower above upper
    with pytest.raises(relp_amd.RelpError):
        relp_amd.Model.from_general_form(columns, [("Range", -1)], b, variables)             # negative range


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_gpu_solves_general_forms_like_the_oracle(seed):
    from relp_oracle import FiniteOptimum, Infeasible, Unbounded, solve_relaxation
    rng = random.Random(7000 + seed)
    form = random_general_form(rng, rng.randint(2, 7), rng.randint(2, 8))
    columns, kinds, b, variables, maximize, fixed_cost = form
    general, data = oracle_standard_form(*form)
    try:
        expected = solve_relaxation(data)
    except AssertionError:
        pytest.skip("the reference's LU cannot factor a 1 x 1 basis")
    model = relp_amd.Model.from_general_form(columns, kinds, b, variables, maximize=maximize, fixed_cost=fixed_cost)
    solver = relp_amd.Solver(certify=1).load_model(model)
    result = solver.solve_relaxation()
    if isinstance(expected, Infeasible):
        assert result.kind == relp_amd.INFEASIBLE
    elif isinstance(expected, Unbounded):
        assert result.kind == relp_amd.UNBOUNDED
    else:
        assert isinstance(expected, FiniteOptimum) and result.kind == relp_amd.FINITE_OPTIMUM and result.certified == 1
        objective = general.objective_of(data.reconstruct_solution(expected.solution))
        assert F(solver.objective_exact()) == objective
    solver.close()

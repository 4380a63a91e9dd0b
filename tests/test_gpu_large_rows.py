"""m > 8192: the ratio test runs across workgroups (k2l_* kernels); this LP also leaves ~8300 artificials basic at level
zero after phase one, so that every one of them goes through the given-row pivot of ``remove_artificial_basis_variables``
(phase_one.rs:232-278) on that path."""
import pytest

import relp_amd

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("implicit_bounds", [0, 1])
def test_zero_level_artificials_are_driven_out_beyond_8192_rows(implicit_bounds):
    m = 8400
    # row i:  x_i - z_i = b_i   (b_i = 1 on every 100th row, else 0);  min sum x_i,  0 <= z_i <= 5  ->  x = b, z = 0
    column_start, rows, nums = [0], [], []
    for i in range(m):      # x_i
        rows.append(i); nums.append(1); column_start.append(len(rows))
    for i in range(m):      # z_i
        rows.append(i); nums.append(-1); column_start.append(len(rows))
    b = [1 if i % 100 == 0 else 0 for i in range(m)]
    cost = [1] * m + [0] * m
    upper = [None] * m + [5] * m
    solver = relp_amd.Solver(certify=0, implicit_bounds=implicit_bounds)
    solver.load_matrix_data(column_start, rows, nums, [1] * len(nums), b=b, cost=cost, upper=upper, counts=(m, 0, 0, 0))
    result = solver.solve_relaxation()
    assert result.kind == relp_amd.FINITE_OPTIMUM
    assert result.objective == pytest.approx(sum(b), abs=1e-9)
    assert result.pivots_phase_one >= m  # one pivot per artificial: 84 by the ratio test, the others with a given row
    x = solver.solution()
    assert all(abs(x[i] - b[i]) <= 1e-9 for i in range(m))
    assert all(abs(x[m + i]) <= 1e-9 for i in range(m))
    solver.close()

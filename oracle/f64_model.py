"""numpy model of the DEVICE f64 algorithm (oracle side; test infrastructure only).

This is the algorithm the HIP kernels in ``relp_amd/csrc`` implement, restated with numpy so that
tolerances and the explicit-inverse / product-form update / Newton-Schulz polish design can be
checked on the CPU and so that GPU results have an f64 twin to be compared with (1e-9 relative on
the objective; pivot sequences may differ in degenerate ties).

Algorithm = the reference's loop (phase_one.rs:134-178, phase_two.rs:36-58) with
``Carry<f64, explicit inverse>``: the basis inverse is the dense matrix the reference's
``BasisInverseRows`` holds sparsely (basis_inverse_rows.rs:21-23), updated by the same row
reduction (basis_inverse_rows.rs:36-70), with tolerances the exact reference does not need.
"""
import numpy as np
import scipy.sparse as sp


class Options:
    tol_dual = 1e-9      # a column is a pricing candidate when cbar_j < -tol_dual
    tol_pivot = 1e-9     # ratio test considers alpha_i > tol_pivot
    harris_delta = 1e-9  # feasibility slack of the Harris ratio test (pass 1)
    tol_zero = 1e-12     # |x| below this counts as zero when driving artificials out
    tol_feas = 1e-7      # phase-one objective above this (relative to 1+|b|_1) => infeasible
    polish_period = 64   # Newton-Schulz polish of the explicit inverse every this many pivots
    max_iters = 200000
    max_seconds = None   # wall-clock budget of `solve` (bench.py's bounded cpu_baseline sample); None: none


OPTIMAL, UNBOUNDED, INFEASIBLE, ITER_LIMIT = "optimal", "unbounded", "infeasible", "iteration_limit"


class Model:
    def __init__(self, provider, options=None):
        self.opt = options or Options()
        self.provider = provider
        m = provider.nr_rows()
        n_p = provider.nr_columns()
        real = provider.pivot_element_indices() if hasattr(provider, "pivot_element_indices") else []
        real_rows = dict(real)
        art_rows = [i for i in range(m) if i not in real_rows]
        self.n_art = len(art_rows)
        self.m, self.n = m, self.n_art + n_p
        rows, cols, vals = [], [], []
        for k, r in enumerate(art_rows):
            rows.append(r); cols.append(k); vals.append(1.0)
        for j in range(n_p):
            for i, v in provider.column(j):
                rows.append(i); cols.append(self.n_art + j); vals.append(float(v))
        self.A = sp.csc_matrix((vals, (rows, cols)), shape=(m, self.n))
        self.AT = self.A.T.tocsr()
        self.cost2 = np.zeros(self.n)
        for j in range(n_p):
            self.cost2[self.n_art + j] = float(provider.cost_value(j))
        self.cost1 = np.zeros(self.n)
        self.cost1[:self.n_art] = 1.0
        self.xB = np.array([float(v) for v in provider.right_hand_side()])
        self.basis = np.zeros(m, dtype=np.int64)
        art_of_row = {r: k for k, r in enumerate(art_rows)}
        for i in range(m):
            self.basis[i] = art_of_row[i] if i in art_of_row else self.n_art + real_rows[i]
        self.pos = -np.ones(self.n, dtype=np.int64)
        self.pos[self.basis] = np.arange(m)
        self.Binv = np.eye(m)
        self.iters = [0, 0]
        self.polishes = 0
        self.max_residual = 0.0

    # ---- pieces that map 1:1 onto kernels ------------------------------------------------------
    def set_phase(self, cost):
        self.cost = cost
        cB = cost[self.basis]
        self.minus_pi = -(cB @ self.Binv)
        self.minus_obj = -float(cB @ self.xB)
        # gamma_j = 1 + ||Binv a_j||^2 (pivot_rule.rs:202-219,299-305)
        G = self.Binv @ self.A
        self.gamma = 1.0 + np.asarray(G.multiply(G).sum(axis=0)).ravel() if sp.issparse(G) \
            else 1.0 + (np.asarray(G) ** 2).sum(axis=0)
        self.pending = None

    def price(self):
        """SE weight update of the previous pivot, then cbar and argmax cbar^2/gamma (last max)."""
        eligible = (self.pos < 0)
        eligible[:self.n_art] = False
        if self.pending is not None:
            rho, w, gamma_q, alpha_pq, q, leaving = self.pending
            abar = self.AT @ rho
            t = self.AT @ w
            g = self.gamma - 2.0 * abar * t + abar * abar * gamma_q
            g = np.maximum(g, 1.0 + abar * abar)
            mask = eligible.copy()
            mask[leaving] = False
            self.gamma = np.where(mask, g, self.gamma)
            self.gamma[leaving] = gamma_q / (alpha_pq * alpha_pq)
            self.pending = None
        cbar = self.cost + self.AT @ self.minus_pi
        cand = eligible & (cbar < -self.opt.tol_dual)
        if not cand.any():
            return None
        key = np.where(cand, cbar * cbar / self.gamma, -1.0)
        best = key.max()
        q = int(np.flatnonzero(key == best)[-1])
        return q, float(cbar[q])

    def ftran(self, q):
        col = self.A[:, q]
        return self.Binv[:, col.indices] @ col.data

    def ratio(self, alpha):
        """Harris two-pass ratio test (f64 only; the exact reference uses tableau/mod.rs:287-313).

        Pass 1 bounds the step with the feasibility slack ``delta``; pass 2 takes, among the rows
        whose ratio does not exceed that bound, the largest pivot (ties: lowest leaving column,
        the reference's Bland rule)."""
        o = self.opt
        idx = np.flatnonzero(alpha > o.tol_pivot)
        if idx.size == 0:
            return None
        a = alpha[idx]
        x = np.maximum(self.xB[idx], 0.0)
        theta_max = ((x + o.harris_delta) / a).min()
        ok = (x / a) <= theta_max
        idx, a = idx[ok], a[ok]
        best = a.max()
        ties = idx[a == best]
        return int(ties[np.argmin(self.basis[ties])])

    def update(self, q, p, alpha, cbar_q):
        alpha_pq = alpha[p]
        w = alpha @ self.Binv                       # old basis (carry/mod.rs:575)
        rho = self.Binv[p, :] / alpha_pq            # new basis row p
        self.Binv -= np.outer(alpha, rho)
        self.Binv[p, :] = rho
        xp = max(self.xB[p], 0.0) / alpha_pq        # carry/mod.rs:295-325 (clamped: Harris may pick xB[p] < 0 by roundoff)
        self.xB -= alpha * xp
        self.xB[p] = xp
        leaving = int(self.basis[p])
        self.basis[p] = q
        self.pos[q] = p
        self.pos[leaving] = -1
        self.minus_pi -= cbar_q * rho               # carry/mod.rs:338-349
        self.minus_obj -= cbar_q * xp
        gamma_q = 1.0 + float(alpha @ alpha)
        self.pending = (rho, w, gamma_q, alpha_pq, q, leaving)

    def polish(self):
        """Newton-Schulz: X <- X + X (I - B X); all GEMM-shaped work (MFMA f64 on the device)."""
        B = self.A[:, self.basis]
        R = np.eye(self.m) - B @ self.Binv
        self.max_residual = max(self.max_residual, float(np.abs(R).max()))
        self.Binv += self.Binv @ R
        # refresh the dependent vectors from the polished inverse
        rhs = np.array([float(v) for v in self.provider.right_hand_side()])
        self.xB = self.Binv @ rhs
        cB = self.cost[self.basis]
        self.minus_pi = -(cB @ self.Binv)
        self.minus_obj = -float(cB @ self.xB)
        self.polishes += 1

    # ---- driver ----------------------------------------------------------------------------------
    def run_phase(self, phase):
        import time
        since = 0
        while True:
            if sum(self.iters) >= self.opt.max_iters:
                return ITER_LIMIT
            if self.opt.max_seconds is not None and time.perf_counter() - self.started > self.opt.max_seconds:
                return ITER_LIMIT
            sel = self.price()
            if sel is None:
                return OPTIMAL
            q, cbar_q = sel
            alpha = self.ftran(q)
            p = self.ratio(alpha)
            if p is None:
                return UNBOUNDED
            self.update(q, p, alpha, cbar_q)
            self.iters[phase - 1] += 1
            since += 1
            if since >= self.opt.polish_period:
                self.polish()
                since = 0

    def drive_out_artificials(self):
        """phase_one.rs:232-278: zero-level pivots on rows whose basic variable is artificial."""
        removed = []
        for r in np.flatnonzero(self.basis < self.n_art):
            row = self.AT @ self.Binv[r, :]
            ok = (self.pos < 0) & (np.abs(row) > 1e-7)
            ok[:self.n_art] = False
            js = np.flatnonzero(ok)
            if js.size == 0:
                removed.append(int(r))
                continue
            q = int(js[0])
            alpha = self.ftran(q)
            cbar_q = float(self.cost[q] + self.AT[q] @ self.minus_pi)
            self.update(q, int(r), alpha, cbar_q)
            self.price()  # applies the pending weight update
            self.iters[0] += 1
        return removed

    def solve(self):
        import time
        self.started = time.perf_counter()
        if self.n_art > 0:
            self.set_phase(self.cost1)
            status = self.run_phase(1)
            if status != OPTIMAL:
                return status
            self.polish()
            if -self.minus_obj > self.opt.tol_feas * (1.0 + np.abs(self.xB).sum()):
                return INFEASIBLE
            self.redundant_rows = self.drive_out_artificials()
        self.set_phase(self.cost2)
        status = self.run_phase(2)
        if status == OPTIMAL:
            self.polish()
        return status

    def objective(self):
        return -self.minus_obj

    def solution(self):
        """Values over the provider's columns (artificials dropped)."""
        x = np.zeros(self.n)
        x[self.basis] = self.xB
        return x[self.n_art:]

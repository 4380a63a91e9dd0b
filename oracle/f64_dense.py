"""numpy f64 simplex for the dense ``A x <= b`` LPs of BASELINE config 3 (oracle side; test infrastructure only).

The same loop as ``oracle/f64_model.py`` (the reference's phase-two loop, phase_two.rs:36-58, entered through the
``FullInitialBasis`` route, two_phase/mod.rs:80-109, with steepest-edge pricing, pivot_rule.rs:221-296, and
``Carry<f64, explicit inverse>``), written for a dense constraint block so that it can run at 4096 x 8192: the matrix is
kept as one (m, n) array and the slack columns stay implicit.  ``bench.py`` times it as the ``cpu_baseline`` of the
dense workloads (numpy's BLAS uses the host's cores; the thread count is reported); ``tests/`` check it against the
committed exact / HiGHS optima of the scaled-down twins.  The exact-rational CPU path is infeasible at this size
(SURVEY.md section 8(d)).
"""
import time

import numpy as np


class DenseModel:
    tol_dual = 1e-9
    tol_pivot = 1e-9
    harris_delta = 1e-9
    polish_period = 512

    def __init__(self, a_columns, b, c):
        """``a_columns``: (n, m) array, row j = structural column j (the layout ``relp_amd.workloads.dense_lp`` returns)."""
        self.A = np.ascontiguousarray(np.asarray(a_columns, dtype=np.float64).T)  # (m, n)
        self.m, self.n = self.A.shape
        self.rhs = np.asarray(b, dtype=np.float64)
        self.cost = np.concatenate([np.asarray(c, dtype=np.float64), np.zeros(self.m)])  # structurals, then slacks
        m, n = self.m, self.n
        self.basis = np.arange(n, n + m)
        self.pos = -np.ones(n + m, dtype=np.int64)
        self.pos[self.basis] = np.arange(m)
        self.Binv = np.eye(m)
        self.xB = self.rhs.copy()
        self.minus_pi = np.zeros(m)
        self.minus_obj = 0.0
        self.gamma = np.concatenate([1.0 + (self.A * self.A).sum(axis=0), np.full(m, 2.0)])  # pivot_rule.rs:299-305 at B = I
        self.pending = None
        self.pivots = 0

    def _times_all_columns(self, vectors):
        """``[A | I]' v`` for each column of ``vectors`` (m, k): one pass over the dense block."""
        return np.concatenate([self.A.T @ vectors, vectors], axis=0)

    def price(self):
        eligible = self.pos < 0
        if self.pending is not None:  # steepest-edge update of the previous pivot fused with this pricing pass
            rho, w, gamma_q, alpha_pq, leaving = self.pending
            products = self._times_all_columns(np.stack([self.minus_pi, rho, w], axis=1))
            cbar = self.cost + products[:, 0]
            abar, t = products[:, 1], products[:, 2]
            g = np.maximum(self.gamma - 2.0 * abar * t + abar * abar * gamma_q, 1.0 + abar * abar)
            mask = eligible.copy()
            mask[leaving] = False
            self.gamma = np.where(mask, g, self.gamma)
            self.gamma[leaving] = gamma_q / (alpha_pq * alpha_pq)
            self.pending = None
        else:
            cbar = self.cost + self._times_all_columns(self.minus_pi[:, None])[:, 0]
        candidates = eligible & (cbar < -self.tol_dual)
        if not candidates.any():
            return None
        key = np.where(candidates, cbar * cbar / self.gamma, -1.0)
        q = int(np.flatnonzero(key == key.max())[-1])  # last maximum (Iterator::max_by_key)
        return q, float(cbar[q])

    def ftran(self, q):
        return self.Binv @ self.A[:, q] if q < self.n else self.Binv[:, q - self.n].copy()

    def ratio(self, alpha):
        idx = np.flatnonzero(alpha > self.tol_pivot)
        if idx.size == 0:
            return None
        a = alpha[idx]
        x = np.maximum(self.xB[idx], 0.0)
        ok = (x / a) <= ((x + self.harris_delta) / a).min()
        idx, a = idx[ok], a[ok]
        ties = idx[a == a.max()]
        return int(ties[np.argmin(self.basis[ties])])

    def update(self, q, p, alpha, cbar_q):
        alpha_pq = alpha[p]
        w = alpha @ self.Binv
        rho = self.Binv[p, :] / alpha_pq
        self.Binv -= np.outer(alpha, rho)
        self.Binv[p, :] = rho
        xp = max(self.xB[p], 0.0) / alpha_pq
        self.xB -= alpha * xp
        self.xB[p] = xp
        leaving = int(self.basis[p])
        self.basis[p] = q
        self.pos[q] = p
        self.pos[leaving] = -1
        self.minus_pi -= cbar_q * rho
        self.minus_obj -= cbar_q * xp
        self.pending = (rho, w, 1.0 + float(alpha @ alpha), alpha_pq, leaving)
        self.pivots += 1

    def polish(self):
        structural = self.basis < self.n
        B = np.zeros((self.m, self.m))
        B[:, structural] = self.A[:, self.basis[structural]]
        slack_positions = np.flatnonzero(~structural)
        B[self.basis[slack_positions] - self.n, slack_positions] = 1.0
        self.Binv += self.Binv @ (np.eye(self.m) - B @ self.Binv)
        self.xB = self.Binv @ self.rhs
        cB = self.cost[self.basis]
        self.minus_pi = -(cB @ self.Binv)
        self.minus_obj = -float(cB @ self.xB)

    def solve(self, max_seconds=None, max_pivots=None):
        """Returns ``"optimal"``, ``"unbounded"`` or ``"limit"``."""
        start = time.perf_counter()
        since = 0
        while True:
            if max_pivots is not None and self.pivots >= max_pivots:
                return "limit"
            if max_seconds is not None and time.perf_counter() - start >= max_seconds:
                return "limit"
            selected = self.price()
            if selected is None:
                self.polish()
                if self.price() is None:
                    return "optimal"
                continue
            q, cbar_q = selected
            alpha = self.ftran(q)
            p = self.ratio(alpha)
            if p is None:
                return "unbounded"
            self.update(q, p, alpha, cbar_q)
            since += 1
            if since >= self.polish_period:
                self.polish()
                since = 0

    def objective(self):
        return -self.minus_obj

"""ctypes binding of ``librelp_amd.so`` (the C ABI in ``include/relp_amd.h``).

This module is plumbing for tests, ``bench.py`` and ``__graft_entry__``: every compute call goes through the C ABI
into the HIP kernels.  There is no Python or CPU implementation behind it -- if the shared library is missing, or no
HIP device is usable, calls raise.  Names mirror the reference's traits (``solve_relaxation``,
``left_multiply_by_basis_inverse`` ...) so that the parity tests read like the reference's own tests.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RELP_AMD_LIB", os.path.join(_HERE, "librelp_amd.so"))  # override: diagnostic builds only

OK, ERR_ARGUMENT, ERR_PARSE, ERR_DEVICE, ERR_OVERFLOW, ERR_STATE, ERR_NUMERICAL = range(7)
FINITE_OPTIMUM, INFEASIBLE, UNBOUNDED, ITERATION_LIMIT = 1, 2, 3, 4
STEEPEST_EDGE, DANTZIG, FIRST_PROFITABLE, FIRST_PROFITABLE_MEMORY = 0, 1, 2, 3
STOP_NO_ENTERING, STOP_UNBOUNDED, STOP_BUDGET = 1, 2, 3
CARRY_EXPLICIT, CARRY_LU, CARRY_LU_INVERSE = 0, 1, 2
RATIO_HARRIS, RATIO_TEXTBOOK, RATIO_AUTO = 0, 1, 2


class RelpError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("relp_amd status %d: %s" % (status, message))
        self.status = status


class Options(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32), ("pivot_rule", C.c_int32), ("polish_period", C.c_int32),
                ("pivots_per_launch", C.c_int32), ("max_pivots", C.c_int64), ("tol_dual", C.c_double),
                ("tol_pivot", C.c_double), ("harris_delta", C.c_double), ("tol_feasible", C.c_double),
                ("certify", C.c_int32), ("use_graph", C.c_int32), ("verbose", C.c_int32), ("implicit_bounds", C.c_int32),
                ("carry", C.c_int32), ("refactor_period", C.c_int32), ("lu_pivot_threshold", C.c_double),
                ("ratio_rule", C.c_int32), ("crash", C.c_int32),
                ("dense_storage", C.c_int32), ("pivot_kernels", C.c_int32), ("product_form", C.c_int32), ("ftran_min_nnz", C.c_int32),
                ("lu_refactor", C.c_int32),
                ("switches", C.c_uint32), ("dense_blocks", C.c_int32), ("ftran_slices", C.c_int32), ("price_lds_max", C.c_int32),
                ("certify_threads", C.c_int32), ("exact_grid", C.c_int32), ("exact_update", C.c_int32), ("luf_dense_tail", C.c_int32),
                ("luf_slack", C.c_int32), ("luf_lds", C.c_int32), ("luf_lds_arena", C.c_int32), ("luf_arena_cap", C.c_int32),
                ("carry_weights_min", C.c_double)]


# relp_switch bits of Options.switches (include/relp_amd.h)
SW_NO_TOUCHED, SW_K2_SINGLE, SW_ELL_WIDE, SW_NO_GENERATED_COLUMNS, SW_NO_SLACK_IN_BTRAN, SW_NO_DENSE_LANE, SW_POLISH_ALWAYS, SW_NO_RHO_BITS, \
    SW_PRICE_UNIT_PAIRS, SW_CERTIFY_NO_LEVELS, SW_GEMM_VECTOR, SW_LUF_CLAIM_TARGETS, SW_LUF_NO_LDS_ARENA, SW_LUI_CLAIM_ROWS, SW_BI_FACTOR_HOST = (1 << k for k in range(15))

# The library reads no environment variable that changes a kernel or a result (round 5).  This BINDING -- test and bench plumbing --
# still maps the old variable names onto option fields for the tools that A/B them; an option given by the caller always wins.
_ENV_SWITCHES = {"RELP_NO_TOUCHED": SW_NO_TOUCHED, "RELP_K2_SINGLE": SW_K2_SINGLE, "RELP_ELL_WIDE": SW_ELL_WIDE,
                 "RELP_NO_GENERATED_COLUMNS": SW_NO_GENERATED_COLUMNS, "RELP_NO_SLACK_IN_BTRAN": SW_NO_SLACK_IN_BTRAN,
                 "RELP_NO_DENSE_LANE": SW_NO_DENSE_LANE, "RELP_POLISH_ALWAYS": SW_POLISH_ALWAYS, "RELP_NO_RHO_BITS": SW_NO_RHO_BITS,
                 "RELP_PRICE_UNIT_PAIRS": SW_PRICE_UNIT_PAIRS, "RELP_CERTIFY_NO_LEVELS": SW_CERTIFY_NO_LEVELS,
                 "RELP_LUF_CLAIM_TARGETS": SW_LUF_CLAIM_TARGETS, "RELP_LUF_NO_LDS_ARENA": SW_LUF_NO_LDS_ARENA,
                 "RELP_LUI_CLAIM_ROWS": SW_LUI_CLAIM_ROWS, "RELP_BI_FACTOR_HOST": SW_BI_FACTOR_HOST}
_ENV_INTEGERS = {"RELP_DENSE_BLOCKS": "dense_blocks", "RELP_FTRAN_SLICES": "ftran_slices", "RELP_FTRAN_MIN_NNZ": "ftran_min_nnz",
                 "RELP_PRICE_LDS_MAX": "price_lds_max", "RELP_CERTIFY_THREADS": "certify_threads", "RELP_EXACT_GRID": "exact_grid",
                 "RELP_EXACT_UPDATE": "exact_update", "RELP_LUF_SLACK": "luf_slack", "RELP_LUF_LDS_ARENA": "luf_lds_arena",
                 "RELP_LUF_ARENA_CAP": "luf_arena_cap"}


def options_from_environment(options, given=()):
    """The old environment hooks onto the fields of `options` that the caller did not set (`given`)."""
    env = os.environ
    if "switches" not in given:
        for name, bit in _ENV_SWITCHES.items():
            if env.get(name):
                options.switches |= bit
        if env.get("RELP_GEMM") == "vector":
            options.switches |= SW_GEMM_VECTOR
    for name, field in _ENV_INTEGERS.items():
        if field not in given and env.get(name):
            setattr(options, field, int(env[name]))
    if "carry_weights_min" not in given and env.get("RELP_CARRY_WEIGHTS_MIN"):
        options.carry_weights_min = float(env["RELP_CARRY_WEIGHTS_MIN"])
    if "luf_dense_tail" not in given and env.get("RELP_LUF_DENSE_TAIL"):
        tail = int(env["RELP_LUF_DENSE_TAIL"])
        options.luf_dense_tail = tail if tail > 0 else -1
    if "luf_lds" not in given and env.get("RELP_LUF_LDS"):
        options.luf_lds = int(env["RELP_LUF_LDS"]) + 1
    if "dense_storage" not in given:
        if env.get("RELP_DENSE_F64"):
            options.dense_storage = 2
        elif env.get("RELP_DENSE_F32"):
            options.dense_storage = 1
    if "pivot_kernels" not in given and env.get("RELP_NO_FUSED"):
        options.pivot_kernels = 1
    if "product_form" not in given and env.get("RELP_ETA") == "0":
        options.product_form = 1
    if "lu_refactor" not in given and env.get("RELP_REFACTOR") in ("device", "host"):
        options.lu_refactor = 1 if env["RELP_REFACTOR"] == "device" else 2
    return options


class Result(C.Structure):
    _fields_ = [("kind", C.c_int32), ("certified", C.c_int32), ("pivots_phase_one", C.c_int64),
                ("pivots_phase_two", C.c_int64), ("polishes", C.c_int64), ("exact_repair_pivots", C.c_int64),
                ("objective", C.c_double), ("solve_seconds", C.c_double), ("certify_seconds", C.c_double),
                ("max_residual", C.c_double), ("refactors", C.c_int64), ("refactor_seconds", C.c_double)]


class ExactResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("limbs", C.c_int32), ("pivots_phase_one", C.c_int64), ("pivots_phase_two", C.c_int64),
                ("trace_entries", C.c_int32), ("objective_length", C.c_int32), ("limbs_tried", C.c_int32 * 6),
                ("redundant_rows", C.c_int32), ("reserved", C.c_int32), ("pivots_survived", C.c_int64 * 6)]


class ExactWidthRecord(C.Structure):
    _fields_ = [("limbs", C.c_int32), ("grid", C.c_int32), ("pivots_total_at_end", C.c_int64), ("seconds", C.c_double),
                ("step_seconds", C.c_double * 10), ("update_word_products_needed", C.c_int64), ("update_word_products_issued", C.c_int64)]


EXACT_STEPS = ("x_B", "reduced costs", "arg-max", "exact weights", "tournament", "entering column", "ratio test", "update of N", "bookkeeping", "products and keys of the candidates")


class BatchEntry(C.Structure):
    _fields_ = [("status", C.c_int32), ("model", C.c_int32), ("worker", C.c_int32), ("device", C.c_int32), ("result", Result),
                ("start_seconds", C.c_double), ("end_seconds", C.c_double)]


class BatchWorker(C.Structure):
    _fields_ = [("device", C.c_int32), ("tickets", C.c_int32), ("pivots", C.c_int64), ("busy_seconds", C.c_double),
                ("queue_seconds", C.c_double), ("idle_seconds", C.c_double), ("finish_seconds", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("launches", C.c_int64), ("price_launches", C.c_int64), ("price_seconds", C.c_double),
                ("update_seconds", C.c_double), ("ftran_seconds", C.c_double), ("price_bytes", C.c_int64),
                ("update_bytes", C.c_int64)]


_lib = None

# every symbol include/relp_amd.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "relp_version", "relp_options_default", "relp_options_default_sized", "relp_model_from_mps", "relp_model_from_mps_ex", "relp_model_from_general_form", "relp_model_from_provider", "relp_model_original_variables", "relp_model_max_flow", "relp_model_shortest_path", "relp_model_free", "relp_model_dimensions",
    "relp_model_column", "relp_model_column_exact", "relp_model_cost", "relp_model_right_hand_side",
    "relp_model_initial_pivots", "relp_model_fixed_cost", "relp_create", "relp_destroy", "relp_last_error",
    "relp_load_matrix_data", "relp_load_dense_le", "relp_load_mps", "relp_load_mps_ex", "relp_get_original_solution", "relp_load_model", "relp_get_dimensions", "relp_get_column",
    "relp_get_cost", "relp_get_right_hand_side", "relp_get_initial_pivots", "relp_solve_relaxation",
    "relp_get_solution", "relp_get_objective_exact", "relp_get_record_json", "relp_solve_exact", "relp_get_exact_counters", "relp_get_basis", "relp_set_basis", "relp_begin_phase_one",
    "relp_begin_phase_two", "relp_bi_ftran", "relp_bi_btran", "relp_bi_row", "relp_price", "relp_relative_costs",
    "relp_get_gamma", "relp_ratio", "relp_bring_into_basis", "relp_get_last_pivot", "relp_se_after_basis_update", "relp_refactor", "relp_iterate", "relp_get_b", "relp_get_objective", "relp_get_stats",
    "relp_reset_stats", "relp_profile_kernel", "relp_debug_stamps", "relp_debug_set_tuning", "relp_debug_exact_finish", "relp_debug_exact_words", "relp_debug_grid_barrier", "relp_debug_exact_tile_bench",
    # BasisInverse as an object of its own (relp_amd/basis_inverse.py)
    "relp_bi_options_default", "relp_bi_identity", "relp_bi_invert", "relp_bi_free", "relp_bi_last_error", "relp_bi_m",
    "relp_bi_left_multiply", "relp_bi_right_multiply", "relp_bi_basis_inverse_row", "relp_bi_generate_element",
    "relp_bi_change_basis", "relp_bi_should_refactor", "relp_bi_remove_basis_part", "relp_bi_statistics",
    "relp_bi_get_factors", "relp_lu_factor_host", "relp_lu_invert_host", "relp_lu_factor_device",
    # BasisInverse over exact rationals (relp_amd/basis_inverse.py: ExactBasisInverse)
    "relp_bix_identity", "relp_bix_invert", "relp_bix_free", "relp_bix_last_error", "relp_bix_m", "relp_bix_result_words", "relp_bix_left_multiply",
    "relp_bix_right_multiply", "relp_bix_right_multiply_words", "relp_bix_basis_inverse_row", "relp_bix_generate_element", "relp_bix_change_basis", "relp_bix_should_refactor", "relp_bix_remove_basis_part",
    # exact solution vector, variable names, batches of independent LPs
    "relp_get_solution_exact", "relp_get_variable_name",
    "relp_batch_create", "relp_batch_destroy", "relp_batch_workers", "relp_batch_run", "relp_batch_get_objective_exact", "relp_batch_handle",
]


def lib():
    """Load the shared library (fails loudly when it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RelpError(ERR_DEVICE, "%s is missing: run __graft_entry__.build() (make -C relp_amd/csrc)" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.relp_version.restype = C.c_char_p
        _lib.relp_last_error.restype = C.c_char_p
        _lib.relp_last_error.argtypes = [C.c_void_p]
    return _lib


def _ptr(array, ctype):
    return array.ctypes.data_as(C.POINTER(ctype))


def default_options(**overrides):
    options = Options()
    status = lib().relp_options_default_sized(C.byref(options), C.sizeof(Options))  # (the caller states the size it was built with)
    if status != OK:
        raise RelpError(status, "relp_options_default_sized refused sizeof(Options) = %d: the binding and the library disagree" % C.sizeof(Options))
    for key, value in overrides.items():
        if not hasattr(options, key):
            raise AttributeError(key)
        setattr(options, key, value)
    return options_from_environment(options, given=overrides)


class Model:
    """Host-only provider (``MatrixData``; matrix_provider/matrix_data.rs:63-102).  Needs no GPU."""

    def __init__(self, path, fixed=None, presolve=False):
        if fixed is None:
            fixed = str(path).upper().endswith(".SIF")  # tests/netlib/mod.rs:55 uses parse_fixed for the .SIF files
        self._h = C.c_void_p()
        error = C.create_string_buffer(512)
        status = lib().relp_model_from_mps_ex(str(path).encode(), int(fixed), int(bool(presolve)), C.byref(self._h), error, 512)
        if status != OK:
            raise RelpError(status, error.value.decode())
        self._read_dimensions()

    def original_variables(self):
        """(number of variables of the file, how many the presolve removed)."""
        total, removed = C.c_int32(), C.c_int32()
        lib().relp_model_original_variables(self._h, C.byref(total), C.byref(removed))
        return total.value, removed.value

    def _read_dimensions(self):
        rows, cols, cons, struct = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        nnz = C.c_int64()
        groups = (C.c_int32 * 4)()
        lib().relp_model_dimensions(self._h, C.byref(rows), C.byref(cols), C.byref(cons), C.byref(struct), C.byref(nnz), groups)
        self.nr_rows, self.nr_columns, self.nr_constraints = rows.value, cols.value, cons.value
        self.nr_structural, self.nnz, self.group_counts = struct.value, nnz.value, list(groups)

    @classmethod
    def _from_graph(cls, entry, nr_vertices, arcs, s, t):
        """``arcs``: iterable of ``(tail, head, value)``; value an int, a ``Fraction`` or a ``(num, den)`` pair.  Sorted
        here by (tail, head): the order the reference's adjacency matrix enumerates them (representation.rs:40-44)."""
        from fractions import Fraction
        arcs = sorted(((int(a), int(b), Fraction(*v) if isinstance(v, tuple) else Fraction(v)) for a, b, v in arcs),
                      key=lambda arc: arc[:2])
        tail = np.array([a for a, _, _ in arcs], dtype=np.int32)
        head = np.array([b for _, b, _ in arcs], dtype=np.int32)
        num = np.array([v.numerator for _, _, v in arcs], dtype=np.int64)
        den = np.array([v.denominator for _, _, v in arcs], dtype=np.int64)
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        error = C.create_string_buffer(512)
        status = getattr(lib(), entry)(int(nr_vertices), len(arcs), _ptr(tail, C.c_int32), _ptr(head, C.c_int32),
                                       _ptr(num, C.c_int64), _ptr(den, C.c_int64), int(s), int(t), C.byref(self._h), error, 512)
        if status != OK:
            raise RelpError(status, error.value.decode())
        self._read_dimensions()
        self.arcs = arcs
        return self

    @classmethod
    def max_flow(cls, nr_vertices, arcs, s, t):
        """The provider of examples/max_flow.rs (`Primal::new`): maximise the flow out of ``s``; value = capacity."""
        return cls._from_graph("relp_model_max_flow", nr_vertices, arcs, s, t)

    @classmethod
    def shortest_path(cls, nr_vertices, arcs, s, t):
        """The provider of examples/shortest_path.rs (`Primal::new`); value = arc length."""
        return cls._from_graph("relp_model_shortest_path", nr_vertices, arcs, s, t)

    @classmethod
    def from_general_form(cls, columns, constraint_types, b, variables, maximize=False, fixed_cost=0, presolve=False):
        """``GeneralForm::new`` (general_form/mod.rs:211-237) + [presolve +] ``standardize`` + ``derive_matrix_data``.

        ``columns``: per variable a list of ``(row, value)`` with ascending rows; ``constraint_types``: per row ``"Equal"``,
        ``"Less"``, ``"Greater"`` or ``("Range", r)``; ``variables``: per variable ``(cost, lower, upper)`` with ``None`` for a
        missing bound.  Numbers: int, ``Fraction`` or ``(num, den)``."""
        from fractions import Fraction

        def pair(v):
            f = Fraction(*v) if isinstance(v, tuple) else Fraction(v)
            return f.numerator, f.denominator
        kinds = {"Equal": 0, "Range": 1, "Less": 2, "Greater": 3}
        n, m = len(columns), len(b)
        start = np.zeros(n + 1, dtype=np.int64)
        rows, v_num, v_den = [], [], []
        for j, column in enumerate(columns):
            for i, value in column:
                rows.append(i)
                a, d = pair(value)
                v_num.append(a)
                v_den.append(d)
            start[j + 1] = len(rows)
        arrays = {
            "rows": np.array(rows or [0], dtype=np.int32), "v_num": np.array(v_num or [0], dtype=np.int64),
            "v_den": np.array(v_den or [1], dtype=np.int64),
            "kind": np.array([kinds[k[0] if isinstance(k, tuple) else k] for k in constraint_types], dtype=np.int32),
            "r_num": np.zeros(m, dtype=np.int64), "r_den": np.ones(m, dtype=np.int64),
            "b_num": np.zeros(m, dtype=np.int64), "b_den": np.ones(m, dtype=np.int64),
            "c_num": np.zeros(n, dtype=np.int64), "c_den": np.ones(n, dtype=np.int64),
            "has_l": np.zeros(n, dtype=np.uint8), "l_num": np.zeros(n, dtype=np.int64), "l_den": np.ones(n, dtype=np.int64),
            "has_u": np.zeros(n, dtype=np.uint8), "u_num": np.zeros(n, dtype=np.int64), "u_den": np.ones(n, dtype=np.int64),
        }
        for i, kind in enumerate(constraint_types):
            if isinstance(kind, tuple):
                arrays["r_num"][i], arrays["r_den"][i] = pair(kind[1])
            arrays["b_num"][i], arrays["b_den"][i] = pair(b[i])
        for j, (cost, lower, upper) in enumerate(variables):
            arrays["c_num"][j], arrays["c_den"][j] = pair(cost)
            if lower is not None:
                arrays["has_l"][j] = 1
                arrays["l_num"][j], arrays["l_den"][j] = pair(lower)
            if upper is not None:
                arrays["has_u"][j] = 1
                arrays["u_num"][j], arrays["u_den"][j] = pair(upper)
        f_num, f_den = pair(fixed_cost)
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        error = C.create_string_buffer(512)
        a = arrays
        status = lib().relp_model_from_general_form(
            int(bool(maximize)), m, n, _ptr(start, C.c_int64), _ptr(a["rows"], C.c_int32), _ptr(a["v_num"], C.c_int64),
            _ptr(a["v_den"], C.c_int64), _ptr(a["kind"], C.c_int32), _ptr(a["r_num"], C.c_int64), _ptr(a["r_den"], C.c_int64),
            _ptr(a["b_num"], C.c_int64), _ptr(a["b_den"], C.c_int64), _ptr(a["c_num"], C.c_int64), _ptr(a["c_den"], C.c_int64),
            _ptr(a["has_l"], C.c_uint8), _ptr(a["l_num"], C.c_int64), _ptr(a["l_den"], C.c_int64),
            _ptr(a["has_u"], C.c_uint8), _ptr(a["u_num"], C.c_int64), _ptr(a["u_den"], C.c_int64),
            C.c_int64(f_num), C.c_int64(f_den), int(bool(presolve)), C.byref(self._h), error, 512)
        if status != OK:
            raise RelpError(status, error.value.decode())
        self._read_dimensions()
        return self

    @classmethod
    def from_provider(cls, provider):
        """Any object with the reference's ``MatrixProvider`` interface (matrix_provider/mod.rs:37-134): ``nr_rows()``,
        ``nr_columns()``, ``column(j)`` -> sorted ``[(row, value)]``, ``cost_value(j)``, ``right_hand_side()`` and, optionally,
        ``pivot_element_indices()`` (``PartialInitialBasis``).  Handed to the C ABI as callbacks (``relp_model_from_provider``)."""
        from fractions import Fraction

        def pair(v):
            f = Fraction(v)
            return f.numerator, f.denominator
        column_t = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64))
        cost_t = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64))
        rhs_t = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64))
        pivots_t = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32))

        class Provider(C.Structure):
            _fields_ = [("user", C.c_void_p), ("nr_rows", C.c_int32), ("nr_columns", C.c_int32), ("column", column_t),
                        ("cost_value", cost_t), ("right_hand_side", rhs_t), ("pivot_element_indices", pivots_t)]

        def column(_, j, capacity, rows, nums, dens):
            entries = provider.column(j)
            for e, (i, v) in enumerate(entries[:capacity]):
                rows[e] = i
                nums[e], dens[e] = pair(v)
            return len(entries)

        def cost_value(_, j, num, den):
            num[0], den[0] = pair(provider.cost_value(j))

        def right_hand_side(_, nums, dens):
            for i, v in enumerate(provider.right_hand_side()):
                nums[i], dens[i] = pair(v)

        def pivot_element_indices(_, capacity, rows, columns):
            pivots = provider.pivot_element_indices()
            for k, (r, c) in enumerate(pivots[:capacity]):
                rows[k], columns[k] = r, c
            return len(pivots)
        record = Provider(None, provider.nr_rows(), provider.nr_columns(), column_t(column), cost_t(cost_value), rhs_t(right_hand_side),
                          pivots_t(pivot_element_indices) if hasattr(provider, "pivot_element_indices") else pivots_t())
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        error = C.create_string_buffer(512)
        status = lib().relp_model_from_provider(C.byref(record), C.byref(self._h), error, 512)
        if status != OK:
            raise RelpError(status, error.value.decode())
        self._read_dimensions()
        return self

    def __del__(self):
        if getattr(self, "_h", None):
            lib().relp_model_free(self._h)
            self._h = None

    def column(self, j):
        count = C.c_int32()
        rows = np.zeros(self.nr_rows, dtype=np.int32)
        vals = np.zeros(self.nr_rows, dtype=np.float64)
        status = lib().relp_model_column(self._h, j, self.nr_rows, C.byref(count), _ptr(rows, C.c_int32), _ptr(vals, C.c_double))
        if status != OK:
            raise RelpError(status, "column")
        return rows[:count.value].copy(), vals[:count.value].copy()

    def column_exact(self, j):
        count = C.c_int32()
        rows = np.zeros(self.nr_rows, dtype=np.int32)
        num = np.zeros(self.nr_rows, dtype=np.int64)
        den = np.zeros(self.nr_rows, dtype=np.int64)
        status = lib().relp_model_column_exact(self._h, j, self.nr_rows, C.byref(count), _ptr(rows, C.c_int32),
                                               _ptr(num, C.c_int64), _ptr(den, C.c_int64))
        if status != OK:
            raise RelpError(status, "column_exact")
        k = count.value
        return [(int(rows[e]), int(num[e]), int(den[e])) for e in range(k)]

    def cost_value(self, j):
        out = C.c_double()
        lib().relp_model_cost(self._h, j, C.byref(out))
        return out.value

    def right_hand_side(self):
        out = np.zeros(self.nr_rows)
        lib().relp_model_right_hand_side(self._h, _ptr(out, C.c_double))
        return out

    def pivot_element_indices(self):
        count = C.c_int32()
        rows = np.zeros(self.nr_rows, dtype=np.int32)
        cols = np.zeros(self.nr_rows, dtype=np.int32)
        lib().relp_model_initial_pivots(self._h, self.nr_rows, C.byref(count), _ptr(rows, C.c_int32), _ptr(cols, C.c_int32))
        return [(int(rows[k]), int(cols[k])) for k in range(count.value)]

    def fixed_cost(self):
        out = C.c_double()
        lib().relp_model_fixed_cost(self._h, C.byref(out))
        return out.value


class Solver:
    """One handle = one LP resident on one GPU (``Tableau<Carry<f64, _>, _>`` + ``PivotRule`` state)."""

    def __init__(self, options=None, **overrides):
        self.options = options or default_options(**overrides)
        self._h = C.c_void_p()
        status = lib().relp_create(C.byref(self.options), C.byref(self._h))
        if status != OK:
            raise RelpError(status, "relp_create failed (no usable HIP device? the product has no CPU fallback)")
        self.m = self.n_provider = self.n_art = 0

    def close(self):
        if getattr(self, "_h", None):
            lib().relp_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, status):
        if status != OK:
            raise RelpError(status, lib().relp_last_error(self._h).decode())

    def _dims(self):
        rows, cols, cons, struct, art = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        nnz = C.c_int64()
        self._check(lib().relp_get_dimensions(self._h, C.byref(rows), C.byref(cols), C.byref(cons), C.byref(struct),
                                              C.byref(art), C.byref(nnz)))
        self.m, self.n_provider, self.n_art, self.nnz = rows.value, cols.value, art.value, nnz.value
        self.n_structural = struct.value
        self.n = self.n_art + self.n_provider

    # ---- provider ------------------------------------------------------------------------------
    def load_mps(self, path, fixed=None, presolve=False):
        """``presolve=True`` applies the reference's ``GeneralForm::presolve`` first (tests/netlib/mod.rs:58)."""
        if fixed is None:
            fixed = str(path).upper().endswith(".SIF")
        self._check(lib().relp_load_mps_ex(self._h, str(path).encode(), int(fixed), int(bool(presolve))))
        self._dims()
        return self

    def load_model(self, model):
        self._check(lib().relp_load_model(self._h, model._h))
        self._dims()
        return self

    def load_dense_le(self, a_column_major, b, cost):
        """Dense `A x <= b` provider (FullInitialBasis route, two_phase/mod.rs:80-109).  `a_column_major`: (n, m) int64."""
        a = np.ascontiguousarray(a_column_major, dtype=np.int64)
        n, m = a.shape
        b = np.ascontiguousarray(b, dtype=np.int64)
        cost = np.ascontiguousarray(cost, dtype=np.int64)
        assert b.shape == (m,) and cost.shape == (n,)
        self._check(lib().relp_load_dense_le(self._h, C.c_int32(m), C.c_int32(n), _ptr(a, C.c_int64), _ptr(b, C.c_int64),
                                             _ptr(cost, C.c_int64)))
        self._dims()
        return self

    def load_matrix_data(self, column_start, row_index, value_num, value_den, b, cost, upper=None, ranges=(),
                         counts=(0, 0, 0, 0), fixed_cost=(0, 1)):
        """``MatrixData::new`` (matrix_data.rs:172-248).  ``b``/``cost``/``upper``/``ranges``: lists of (num, den) or ints."""
        def pairs(values):
            num = np.array([v[0] if isinstance(v, tuple) else v for v in values], dtype=np.int64)
            den = np.array([v[1] if isinstance(v, tuple) else 1 for v in values], dtype=np.int64)
            return num, den
        column_start = np.ascontiguousarray(column_start, dtype=np.int64)
        row_index = np.ascontiguousarray(row_index, dtype=np.int32)
        value_num = np.ascontiguousarray(value_num, dtype=np.int64)
        value_den = np.ascontiguousarray(value_den, dtype=np.int64)
        n = len(column_start) - 1
        b_num, b_den = pairs(b)
        c_num, c_den = pairs(cost)
        has_upper = np.zeros(n, dtype=np.uint8)
        u_num = np.zeros(n, dtype=np.int64)
        u_den = np.ones(n, dtype=np.int64)
        if upper is not None:
            for j, u in enumerate(upper):
                if u is not None:
                    has_upper[j] = 1
                    u_num[j], u_den[j] = (u if isinstance(u, tuple) else (u, 1))
        r_num, r_den = pairs(list(ranges)) if len(ranges) else (np.zeros(1, dtype=np.int64), np.ones(1, dtype=np.int64))
        self._keep = (column_start, row_index, value_num, value_den, b_num, b_den, c_num, c_den, has_upper, u_num, u_den, r_num, r_den)
        self._check(lib().relp_load_matrix_data(
            self._h, C.c_int32(len(b)), C.c_int32(n), _ptr(column_start, C.c_int64), _ptr(row_index, C.c_int32),
            _ptr(value_num, C.c_int64), _ptr(value_den, C.c_int64), _ptr(b_num, C.c_int64), _ptr(b_den, C.c_int64),
            _ptr(c_num, C.c_int64), _ptr(c_den, C.c_int64), _ptr(has_upper, C.c_uint8), _ptr(u_num, C.c_int64),
            _ptr(u_den, C.c_int64), _ptr(r_num, C.c_int64), _ptr(r_den, C.c_int64),
            C.c_int32(counts[0]), C.c_int32(counts[1]), C.c_int32(counts[2]), C.c_int32(counts[3]),
            C.c_int64(fixed_cost[0]), C.c_int64(fixed_cost[1])))
        self._dims()
        return self

    # ---- solve_relaxation ------------------------------------------------------------------------
    def solve_relaxation(self):
        result = Result()
        self._check(lib().relp_solve_relaxation(self._h, C.byref(result)))
        return result

    def solution(self):
        """``FiniteOptimum`` vector after ``reconstruct_solution`` (structural columns only)."""
        out = np.zeros(self.n_structural)
        self._check(lib().relp_get_solution(self._h, _ptr(out, C.c_double)))
        return out

    def original_solution(self):
        """Values of the file's variables (shifts, flips, free splits and presolve removals undone), in file order."""
        count = C.c_int32()
        lib().relp_get_original_solution(self._h, 0, None, C.byref(count))
        out = np.zeros(count.value)
        if count.value:
            self._check(lib().relp_get_original_solution(self._h, count.value, _ptr(out, C.c_double), C.byref(count)))
        return out

    def solve_exact(self, first_limbs=2, max_limbs=32, max_pivots=0, trace_capacity=1 << 16):
        """The loop in exact fixed-width integer arithmetic on the device (``relp_solve_exact``).  Returns a dict: ``status``
        (1 optimal, 2 infeasible, 3 unbounded, 4 overflow, 5 pivot limit, 6 redundant rows), ``limbs``, pivots per phase,
        ``trace`` = [(phase, q, p, leaving)], ``objective`` "num/den", ``basis`` and ``survived`` = [(limbs, pivots)] per width tried."""
        result = ExactResult()
        trace = np.zeros(4 * max(1, trace_capacity), dtype=np.int32)
        objective = C.create_string_buffer(1 << 16)
        basis = np.zeros(self.m, dtype=np.int32)
        self._check(lib().relp_solve_exact(self._h, int(first_limbs), int(max_limbs), C.c_int64(int(max_pivots)), C.byref(result),
                                           int(trace_capacity), _ptr(trace, C.c_int32), objective, len(objective), _ptr(basis, C.c_int32)))
        entries = result.trace_entries
        return {"status": result.status, "limbs": result.limbs, "pivots_phase_one": result.pivots_phase_one,
                "pivots_phase_two": result.pivots_phase_two,
                "trace": [tuple(int(v) for v in trace[4 * k:4 * k + 4]) for k in range(entries)],
                "objective": objective.value.decode(), "basis": basis, "redundant_rows": int(result.redundant_rows),
                "survived": [(int(result.limbs_tried[k]), int(result.pivots_survived[k])) for k in range(6) if result.limbs_tried[k]]}

    def exact_counters(self):
        """Per width tried by the last ``solve_exact``: seconds, seconds per step of the loop, and the word products (64 x 64 -> 128 bit)
        of the update of N, needed and issued (``relp_get_exact_counters``)."""
        records = (ExactWidthRecord * 16)()
        count = C.c_int32()
        self._check(lib().relp_get_exact_counters(self._h, records, 16, C.byref(count)))
        return [{"limbs": r.limbs, "grid": r.grid, "pivots_total_at_end": r.pivots_total_at_end, "seconds": r.seconds,
                 "step_seconds": dict(zip(EXACT_STEPS, list(r.step_seconds))),
                 "update_word_products_needed": r.update_word_products_needed, "update_word_products_issued": r.update_word_products_issued}
                for r in records[:min(16, count.value)]]

    def record(self):
        """The per-LP record of the last solve as a dict (``relp_get_record_json``)."""
        import json
        length = C.c_int32()
        self._check(lib().relp_get_record_json(self._h, None, 0, C.byref(length)))
        buf = C.create_string_buffer(length.value + 1)
        self._check(lib().relp_get_record_json(self._h, buf, length.value + 1, C.byref(length)))
        return json.loads(buf.value.decode())

    def objective_exact(self):
        length = C.c_int32()
        lib().relp_get_objective_exact(self._h, None, 0, C.byref(length))
        buf = C.create_string_buffer(length.value + 1)
        self._check(lib().relp_get_objective_exact(self._h, buf, length.value + 1, C.byref(length)))
        return buf.value.decode()

    def solution_exact(self, original=False):
        """``OptimizationResult::FiniteOptimum(SparseVector<RationalBig>)`` in exact form (``relp_get_solution_exact``): a dict
        index -> ``Fraction``.  ``original=False``: structural columns of the standard form after ``reconstruct_solution``;
        ``original=True``: the variables of the file (``compute_full_solution_with_reduced_solution``), by file order."""
        from fractions import Fraction
        count, length = C.c_int32(), C.c_int64()
        self._check(lib().relp_get_solution_exact(self._h, int(bool(original)), 0, C.byref(count), None, None, C.c_int64(0), C.byref(length)))
        if count.value == 0:
            return {}
        index = np.zeros(count.value, dtype=np.int32)
        buf = C.create_string_buffer(length.value)
        self._check(lib().relp_get_solution_exact(self._h, int(bool(original)), count.value, C.byref(count), _ptr(index, C.c_int32), buf,
                                                  C.c_int64(length.value), C.byref(length)))
        texts = buf.value.decode().split("\n")
        return {int(index[k]): Fraction(texts[k]) for k in range(count.value)}

    def variable_name(self, j):
        length = C.c_int32()
        self._check(lib().relp_get_variable_name(self._h, int(j), None, 0, C.byref(length)))
        buf = C.create_string_buffer(length.value + 1)
        self._check(lib().relp_get_variable_name(self._h, int(j), buf, length.value + 1, C.byref(length)))
        return buf.value.decode()

    def basis(self):
        out = np.zeros(self.m, dtype=np.int32)
        self._check(lib().relp_get_basis(self._h, _ptr(out, C.c_int32)))
        return out

    def set_basis(self, basis):
        arr = np.ascontiguousarray(basis, dtype=np.int32)
        self._check(lib().relp_set_basis(self._h, _ptr(arr, C.c_int32)))

    # ---- fine-grained trait ops -------------------------------------------------------------------
    def begin_phase_one(self):
        self._check(lib().relp_begin_phase_one(self._h))

    def begin_phase_two(self):
        self._check(lib().relp_begin_phase_two(self._h))

    def left_multiply_by_basis_inverse(self, rows, values):
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        values = np.ascontiguousarray(values, dtype=np.float64)
        out = np.zeros(self.m)
        self._check(lib().relp_bi_ftran(self._h, len(rows), _ptr(rows, C.c_int32), _ptr(values, C.c_double), _ptr(out, C.c_double)))
        return out

    def right_multiply_by_basis_inverse(self, rows, values):
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        values = np.ascontiguousarray(values, dtype=np.float64)
        out = np.zeros(self.m)
        self._check(lib().relp_bi_btran(self._h, len(rows), _ptr(rows, C.c_int32), _ptr(values, C.c_double), _ptr(out, C.c_double)))
        return out

    def basis_inverse_row(self, row):
        out = np.zeros(self.m)
        self._check(lib().relp_bi_row(self._h, int(row), _ptr(out, C.c_double)))
        return out

    def select_primal_pivot_column(self):
        column, cost = C.c_int32(), C.c_double()
        self._check(lib().relp_price(self._h, C.byref(column), C.byref(cost)))
        return (None if column.value < 0 else (column.value, cost.value))

    def relative_costs(self):
        out = np.zeros(self.n)
        self._check(lib().relp_relative_costs(self._h, _ptr(out, C.c_double)))
        return out

    def gamma(self):
        out = np.zeros(self.n)
        self._check(lib().relp_get_gamma(self._h, _ptr(out, C.c_double)))
        return out

    def select_primal_pivot_row(self, column):
        row = C.c_int32()
        alpha = np.zeros(self.m)
        self._check(lib().relp_ratio(self._h, int(column), C.byref(row), _ptr(alpha, C.c_double)))
        return (None if row.value < 0 else row.value), alpha

    def bring_into_basis(self, column, row):
        """``Tableau::bring_into_basis`` with a given pivot (index space of ``select_primal_pivot_column``)."""
        self._check(lib().relp_bring_into_basis(self._h, int(column), int(row)))

    def last_pivot(self):
        """``BasisChangeComputationInfo`` of the last pivot: ``(phase, pivot_column, pivot_row, leaving_column)``."""
        phase, q, p, leaving = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        self._check(lib().relp_get_last_pivot(self._h, C.byref(phase), C.byref(q), C.byref(p), C.byref(leaving)))
        return phase.value, q.value, p.value, leaving.value

    def after_basis_update(self):
        """``PivotRule::after_basis_update`` (strategy/pivot_rule.rs:243-296): apply the pending steepest-edge weight update."""
        self._check(lib().relp_se_after_basis_update(self._h))

    def refactor(self):
        """Polish the resident inverse now; returns the residual max|I - B'T| found before."""
        out = C.c_double()
        self._check(lib().relp_refactor(self._h, C.byref(out)))
        return out.value

    def iterate(self, count):
        done, reason = C.c_int64(), C.c_int32()
        self._check(lib().relp_iterate(self._h, int(count), C.byref(done), C.byref(reason)))
        return done.value, reason.value

    def b(self):
        out = np.zeros(self.m)
        self._check(lib().relp_get_b(self._h, _ptr(out, C.c_double)))
        return out

    def objective_function_value(self):
        out = C.c_double()
        self._check(lib().relp_get_objective(self._h, C.byref(out)))
        return out.value

    def stats(self):
        stats = Stats()
        self._check(lib().relp_get_stats(self._h, C.byref(stats)))
        return stats

    def debug_stamps(self):
        out = np.zeros(64, dtype=np.uint64)
        self._check(lib().relp_debug_stamps(self._h, _ptr(out, C.c_uint64)))
        return out

    def profile_kernel(self, which, repetitions):
        """Average duration (seconds) of one launch of kernel ``which`` (0 price, 1 ftran+ratio (dry), 2 update)
        measured with HIP events on the handle's stream around ``repetitions`` back-to-back launches."""
        out = C.c_double()
        self._check(lib().relp_profile_kernel(self._h, int(which), int(repetitions), C.byref(out)))
        return out.value


class Batch:
    """``relp_batch_*``: independent LPs resident on the workers of one or more devices, served from one ticket queue by host
    threads inside the library.  ``models``: ``Model`` objects (kept alive here)."""

    TICKET_FN = C.CFUNCTYPE(C.c_int64, C.c_void_p)

    def __init__(self, models, devices=(0,), workers_per_device=1, options=None, **overrides):
        self.models = list(models)
        self.options = options or default_options(**overrides)
        self._h = C.c_void_p()
        handles = (C.c_void_p * len(self.models))(*[m._h for m in self.models])
        devs = np.ascontiguousarray(list(devices), dtype=np.int32)
        error = C.create_string_buffer(512)
        status = lib().relp_batch_create(handles, len(self.models), C.byref(self.options), _ptr(devs, C.c_int32), len(devs),
                                         int(workers_per_device), C.byref(self._h), error, 512)
        if status != OK:
            self._h = None
            raise RelpError(status, error.value.decode() or "relp_batch_create failed")
        n = C.c_int32()
        lib().relp_batch_workers(self._h, C.byref(n))
        self.n_workers = n.value

    def close(self):
        if getattr(self, "_h", None):
            lib().relp_batch_destroy(self._h)
            self._h = None

    __del__ = close

    def run(self, schedule, next_ticket=None):
        """Serve the tickets ``0 .. len(schedule)-1`` (ticket t = model ``schedule[t]``).  ``next_ticket``: optional callable
        returning a fresh ticket per call (a queue shared with other processes).  Returns (entries, workers, makespan)."""
        sched = np.ascontiguousarray(schedule, dtype=np.int32)
        entries = (BatchEntry * len(sched))()
        workers = (BatchWorker * self.n_workers)()
        makespan = C.c_double()
        failure = []

        def draw(_user):  # (an exception inside a ctypes callback would be printed and turned into ticket 0)
            try:
                return int(next_ticket())
            except BaseException as error:  # noqa: BLE001
                failure.append(error)
                return -1

        callback = self.TICKET_FN(draw) if next_ticket is not None else None
        fn = lib().relp_batch_run
        fn.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        status = fn(self._h, _ptr(sched, C.c_int32), len(sched), C.cast(callback, C.c_void_p) if callback else None, None,
                    C.cast(entries, C.c_void_p), C.cast(workers, C.c_void_p), C.byref(makespan))
        if failure:
            raise failure[0]
        if status != OK:
            raise RelpError(status, "relp_batch_run (a ticket handed out twice?)")
        return list(entries), list(workers), makespan.value

    def objective_exact(self, ticket):
        length = C.c_int32()
        if lib().relp_batch_get_objective_exact(self._h, C.c_int64(ticket), None, 0, C.byref(length)) != OK:
            return None
        buf = C.create_string_buffer(length.value + 1)
        lib().relp_batch_get_objective_exact(self._h, C.c_int64(ticket), buf, length.value + 1, C.byref(length))
        return buf.value.decode()

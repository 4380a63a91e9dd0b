/*
 * relp_amd -- C ABI of the MI355X-native revised-simplex hot path (drop-in boundary).
 *
 * The reference (vandenheuvel/relp, Rust) has no FFI today; its plug-in surface for this path is four generic
 * traits.  Each entry point below names the reference interface it replaces; paths are
 * relative to /root/reference/src/algorithm/two_phase/.  A Rust shim that binds these is shown in INTEGRATION.md.
 *
 * Conventions: every call returns a relp_status (0 = ok); arrays are caller-owned; indices are 0-based;
 * a handle is confined to one host thread and owns one HIP stream on one device; distinct handles are independent.
 * Provider column indices are the reference's `MatrixData` indices (matrix_data.rs:115-145): structural columns
 * first, then the five virtual slack groups.  Exact values cross the boundary as (numerator, denominator) pairs of
 * int64 -- the reference's `Rational64` input type (io/mps/number/parse.rs:46-65).
 */
#ifndef RELP_AMD_H
#define RELP_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct relp_handle relp_handle;

typedef enum relp_status {
    RELP_OK = 0,
    RELP_ERR_ARGUMENT = 1,      /* bad pointer / size / index */
    RELP_ERR_PARSE = 2,         /* io::error::{Parse, Inconsistency} (src/io/error.rs:15-237) */
    RELP_ERR_DEVICE = 3,        /* no HIP device or a HIP call failed: the product has NO CPU fallback */
    RELP_ERR_OVERFLOW = 4,      /* fixed-width exact arithmetic overflowed */
    RELP_ERR_STATE = 5,         /* call out of order (e.g. solve before load) */
    RELP_ERR_NUMERICAL = 6      /* f64 breakdown that polishing could not repair */
} relp_status;

/* algorithm/mod.rs:43-47 `OptimizationResult` plus the states a library must report instead of panicking. */
typedef enum relp_result_kind {
    RELP_RESULT_NONE = 0,
    RELP_RESULT_FINITE_OPTIMUM = 1,
    RELP_RESULT_INFEASIBLE = 2,
    RELP_RESULT_UNBOUNDED = 3,
    RELP_RESULT_ITERATION_LIMIT = 4
} relp_result_kind;

/* strategy/pivot_rule.rs:86-305: the four `PivotRule` implementations. */
typedef enum relp_pivot_rule {
    RELP_PIVOT_STEEPEST_EDGE = 0,          /* SteepestDescentAlongObjective (the reference's hard-wired default) */
    RELP_PIVOT_DANTZIG = 1,                /* SteepestDescentAlongVariable */
    RELP_PIVOT_FIRST_PROFITABLE = 2,       /* FirstProfitable */
    RELP_PIVOT_FIRST_PROFITABLE_MEMORY = 3 /* FirstProfitableWithMemory */
} relp_pivot_rule;

/* The two `BasisInverse` implementations of the reference (carry/mod.rs:69-169), both resident on the device. */
typedef enum relp_carry {
    RELP_CARRY_EXPLICIT = 0,   /* `BasisInverseRows` (carry/basis_inverse_rows.rs:21-229): explicit inverse, product-form update */
    RELP_CARRY_LU = 1,         /* `LUDecomposition` (carry/lower_upper/mod.rs:36-272): P B Q = L U, Forrest-Tomlin updates,
                                  refactorisation every `refactor_period` updates */
    RELP_CARRY_LU_INVERSE = 2  /* the same factorisation (`LUDecomposition::invert`, lower_upper/mod.rs:78-92, at every
                                  refactorisation) applied through the sparse INVERSES of its two triangles: FTRAN / BTRAN
                                  (mod.rs:180-237) are two sparse matrix-vector products each instead of two level-by-level
                                  triangular solves, and the basis changes between refactorisations are kept in product form on
                                  top (the etas of `BasisInverseRows`, basis_inverse_rows.rs:96-135, as at most `refactor_period`
                                  columns) instead of Forrest-Tomlin row etas.  At most about 5800 rows (four vectors in LDS up to
                                  about 4300 rows, three beyond). */
} relp_carry;

/* `Tableau::select_primal_pivot_row` (tableau/mod.rs:287-313). */
typedef enum relp_ratio_rule {
    RELP_RATIO_HARRIS = 0,     /* two-pass Harris test with slack `harris_delta`, largest pivot among the near-ties (what f64
                                  needs on real data), ties to the lowest leaving column */
    RELP_RATIO_TEXTBOOK = 1,   /* the reference's rule: the exact minimum ratio, ties to the lowest leaving column (Bland).  For
                                  data on which f64 is exact (small integers); rows <= 8192 (RELP_ERR_ARGUMENT at load beyond:
                                  the multi-workgroup ratio test implements the two-pass rule only) */
    RELP_RATIO_AUTO = 2        /* the default since round 6 ("defaults reproduce the reference"): the
                                  reference's rule where the data are small integers -- every matrix entry an integer of at most
                                  64 in magnitude, every cost and right-hand side an integer below 2^20 -- and the LP has at most
                                  8192 rows; the two-pass Harris test otherwise (decimal data: Netlib).  The per-LP record
                                  (relp_get_record_json) says which one ran */
} relp_ratio_rule;

/* Storage type of a dense column block (`relp_options.dense_storage`). */
typedef enum relp_dense_storage {
    RELP_DENSE_NARROWEST = 0,  /* the narrowest type that holds every entry exactly: signed bytes, else float, else double */
    RELP_DENSE_FLOAT = 1,      /* float when every entry is exactly representable, else double */
    RELP_DENSE_DOUBLE = 2      /* double (SURVEY.md section 8(d)'s 8 bytes per entry) */
} relp_dense_storage;

/* Where the LU carries refactorise (`relp_options.lu_refactor`). */
typedef enum relp_lu_refactor {
    RELP_REFACTOR_AUTO = 0,    /* the faster of the two at the sizes measured so far: today the host path for every shipped LP
                                  (DESIGN.md section 2d has the table: the kernels are 7-25 % behind the host path per pivot) */
    RELP_REFACTOR_DEVICE = 1,  /* kernels on the handle's stream (lu_factor.hip, lu_device_tasks.hip): Markowitz factorisation with
                                  independent pivots per round, inversion of the two triangles, slot records -- no basis read-back, no
                                  upload; RELP_CARRY_LU_INVERSE only (the Forrest-Tomlin carry factorises on the host); what the
                                  kernels cannot take (a row of more than 256 entries, a capacity) falls back to the host path */
    RELP_REFACTOR_HOST = 2,    /* one host core: Markowitz + inversion + task lists, one upload (rounds 2-3) */
    RELP_REFACTOR_DEVICE_ASYNC = 3  /* round 5: the same kernels on a SECOND stream, into a second set of factor arrays, while the
                                  pivots go on with the old factors: the pivot kernel logs the row factors of its etas from the
                                  moment the basis is snapshotted, and when the new factors are ready a replay kernel folds the
                                  logged etas (at most 40) into their product form -- B_now^-1 = E_k ... E_1 B_snapshot^-1 -- and the
                                  handle swaps sets.  RELP_CARRY_LU_INVERSE with its four vectors in LDS (m <= ~4300); anything
                                  else, and any refactorisation the kernels give up on, takes the synchronous paths above */
} relp_lu_refactor;

/* A/B switches of the kernels (`relp_options.switches`; tests and measurements -- all clear = what the library would choose).
 * Round 5: these were environment variables of the same names (RELP_NO_TOUCHED ...); the library no longer reads the environment
 * for anything that changes a kernel or a result, so a caller's options cannot be overruled from outside. */
typedef enum relp_switch {
    RELP_SW_NO_TOUCHED = 1 << 0,            /* no bookkeeping of the non-unit columns of the inverse (m > 2048) */
    RELP_SW_K2_SINGLE = 1 << 1,             /* the one-workgroup ratio test beyond 8192 rows as well */
    RELP_SW_ELL_WIDE = 1 << 2,              /* graph LPs: the 8-wide padded copy of the columns */
    RELP_SW_NO_GENERATED_COLUMNS = 1 << 3,  /* graph LPs: materialised instead of generated incidence columns */
    RELP_SW_NO_SLACK_IN_BTRAN = 1 << 4,     /* dense pipeline: slack columns priced by a launch of their own */
    RELP_SW_NO_DENSE_LANE = 1 << 5,         /* dense block priced by one wave per column instead of a column per lane */
    RELP_SW_POLISH_ALWAYS = 1 << 6,         /* polish at every opportunity */
    RELP_SW_NO_RHO_BITS = 1 << 7,           /* generated columns: no bit table of rho_p's rows */
    RELP_SW_PRICE_UNIT_PAIRS = 1 << 8,      /* generated columns: the pair-per-lane pricing kernel */
    RELP_SW_CERTIFY_NO_LEVELS = 1 << 9,     /* exact certificate: modular inverse without level scheduling */
    RELP_SW_GEMM_VECTOR = 1 << 10,          /* polish: the plain-FMA GEMM instead of v_mfma_f64_16x16x4_f64 */
    RELP_SW_LUF_CLAIM_TARGETS = 1 << 11,    /* device refactorisation: target rows claimed from a counter */
    RELP_SW_LUF_NO_LDS_ARENA = 1 << 12,     /* ... the active sub-matrix in global memory throughout */
    RELP_SW_LUI_CLAIM_ROWS = 1 << 13,       /* ... inversion of the triangles: rows claimed from a counter */
    RELP_SW_BI_FACTOR_HOST = 1 << 14        /* stand-alone BasisInverse: `invert` on a host core */
} relp_switch;

typedef struct relp_options {
    int32_t struct_size;       /* sizeof(relp_options) as the CALLER compiled it; relp_options_default() -- through the header's
                                  macro relp_options_default_sized(o, sizeof(relp_options)) -- sets it and is the only supported
                                  way to initialise the struct.  relp_create / relp_batch_create read that many bytes and take the
                                  defaults for the fields a caller built against an older header does not have; anything but
                                  a size this struct has had in some round is RELP_ERR_ARGUMENT */
    int32_t device;            /* HIP device ordinal */
    int32_t pivot_rule;        /* relp_pivot_rule */
    int32_t polish_period;     /* pivots between Newton-Schulz polishes of the explicit inverse (role of
                                  BasisInverse::should_refactor, lower_upper/mod.rs:249-252) */
    int32_t pivots_per_launch; /* pivots enqueued per host round trip (hipGraph replay length) */
    int64_t max_pivots;        /* iteration cap (the reference has none; cycling is acknowledged, tests/netlib/test.rs:221);
                                  0 = 200 (m + n) + 100 000, after which the result is RELP_RESULT_ITERATION_LIMIT */
    double tol_dual;           /* cbar_j < -tol_dual makes j a pricing candidate */
    double tol_pivot;          /* alpha_i > tol_pivot takes part in the ratio test */
    double harris_delta;       /* feasibility slack of the two-pass ratio test */
    double tol_feasible;       /* phase-one objective above tol_feasible*(1+|b|_1) => infeasible */
    int32_t certify;           /* 1: after the f64 solve, prove the basis optimal in exact arithmetic and return the
                                  exact rational objective (bit-exact with the reference's RationalBig optimum) */
    int32_t use_graph;         /* 1: replay the pivot loop from a hipGraph */
    int32_t verbose;
    int32_t implicit_bounds;   /* 1: variable upper bounds (MatrixData's VariableBound / SlackBound rows, matrix_data.rs:104-112)
                                  are handled by the bounded-variable ratio test instead of as rows: same optimum, fewer rows.
                                  relp_get_basis / relp_set_basis keep speaking the reference's formulation (one column per
                                  row of MatrixData, bound rows included); the other fine-grained trait operations refer to
                                  the reduced LP */
    int32_t carry;             /* relp_carry: which `BasisInverse` the loop maintains (the `BI` of `Carry<F, BI>`,
                                  tests/netlib/mod.rs:62) */
    int32_t refactor_period;   /* LU carry: Forrest-Tomlin updates between refactorisations (`should_refactor`,
                                  lower_upper/mod.rs:249-252: the reference refactors after 31); at most 63; 0 = the default:
                                  31, and 47 product-form updates for RELP_CARRY_LU_INVERSE */
    double lu_pivot_threshold; /* LU carry: relative pivot tolerance of the Markowitz factorisation (f64 needs one, the exact
                                  reference does not); 0 = 0.1 */
    int32_t ratio_rule;        /* relp_ratio_rule */
    int32_t crash;             /* 1: phase one starts from a triangular crash basis instead of one artificial per row without a
                                  slack pivot (an extension; the reference has none, kind/artificial/partially.rs:125-205):
                                  columns with a single entry in the rows still on an artificial are assigned breadth first
                                  -- a spanning forest on the graph providers -- and the crash is kept when it is primal
                                  feasible.  Same optimum, different (much shorter) pivot sequence; explicit carry only */
    /* ---- appended in round 4: what was reachable only through the environment before ---- */
    int32_t dense_storage;     /* relp_dense_storage: storage type of a dense column block (relp_load_dense_le; BASELINE config 3
                                  names the f64 block) */
    int32_t pivot_kernels;     /* 0 = automatic (ratio test and inverse update fused into one launch for m <= 2048, explicit
                                  carry), 1 = the three separate kernels (bit-identical results) */
    int32_t product_form;      /* dense pipeline: 0 = automatic (updates deferred in product form, one rank-k update every 32
                                  pivots), 1 = a rank-one update of the stored inverse per pivot */
    int32_t ftran_min_nnz;     /* columns longer than this take the multi-block FTRAN pipeline; 0 = 1024 */
    int32_t lu_refactor;       /* relp_lu_refactor: where `BasisInverse::invert` of the LU carries runs (lower_upper/mod.rs:78-92,
                                  decomposition/mod.rs:27-143) */
    /* ---- appended in round 5: the remaining A/B hooks (0 = the library's choice everywhere) ---- */
    uint32_t switches;         /* relp_switch bits */
    int32_t dense_blocks;      /* workgroups of the one-wave-per-column dense pricing kernel (256) */
    int32_t ftran_slices;      /* slices of the multi-block FTRAN (by the longest column, at most 64) */
    int32_t price_lds_max;     /* pricing stages -pi / rho / w in LDS up to this many bytes of the three vectors (96 KB) */
    int32_t certify_threads;   /* host threads of the exact certificate (by the core count, at most 32) */
    int32_t exact_grid;        /* relp_solve_exact: workgroups of the cooperative launch (by the work of a pivot) */
    int32_t exact_update;      /* relp_solve_exact, bits: 1 = the update of N = D B^-1 on the vector unit only (default: on the matrix
                                  cores from 16 limbs on); 4 = the update on the matrix cores in the two passes of round 5 (numerators,
                                  then carries / shift / sign by a thread per entry; default since round 6: the tiles finish their
                                  entries themselves into a second buffer of N); 2 = the pricing pass forms the products N a_j of every column that can enter
                                  exactly (default: weight estimates from the leading words, exact only where their error bound asks) */
    int32_t luf_dense_tail;    /* device refactorisation: rows of the dense tail (8); -1 = none */
    int32_t luf_slack;         /* ... Markowitz score slack of a round (16) */
    int32_t luf_lds;           /* ... 1 + the LDS level forced (1: every work array in global memory) */
    int32_t luf_lds_arena;     /* ... cap on the entries of the active sub-matrix held in LDS (test hook: spills) */
    int32_t luf_arena_cap;     /* ... cap on the words of the global arena (test hook: a basis that outgrows it) */
    double carry_weights_min;  /* steepest-edge weights carried from phase one into phase two from this many columns (4e9) */
} relp_options;

typedef struct relp_result {
    int32_t kind;              /* relp_result_kind */
    int32_t certified;         /* 1 when the exact certificate holds */
    int64_t pivots_phase_one;  /* bring_into_basis calls: phase_one.rs:145,265 */
    int64_t pivots_phase_two;  /* phase_two.rs:47 */
    int64_t polishes;
    int64_t exact_repair_pivots;
    double objective;          /* f64 objective incl. fixed cost (general_form/mod.rs:840-851) */
    double solve_seconds;      /* wall clock of solve_relaxation only (device-resident in, result out) */
    double certify_seconds;
    double max_residual;       /* largest |I - B Binv| entry seen by a polish */
    int64_t refactors;         /* LU carry: refactorisations (`BasisInverse::invert`) made, and the host time they took */
    double refactor_seconds;
} relp_result;

typedef struct relp_stats {
    int64_t launches;          /* kernel launches issued */
    int64_t price_launches;
    double price_seconds;      /* HIP-event time spent in the pricing kernel (the dominant kernel), see bench.py */
    double update_seconds;
    double ftran_seconds;
    int64_t price_bytes;       /* algorithmic bytes of one pricing launch */
    int64_t update_bytes;
} relp_stats;

/* The defaults, written into a struct of `caller_size` bytes -- sizeof(relp_options) as the CALLER compiled it: not a byte more is
 * touched, and struct_size is set to that size, so a program built against an older, shorter header keeps working with a newer
 * library (relp_create / relp_batch_create read struct_size bytes and take the defaults for the rest).  The only supported
 * initialiser; a size no header of this library ever had is RELP_ERR_ARGUMENT.  Bindings by symbol (ctypes, Rust `extern`)
 * call this one with their own struct's size. */
int32_t relp_options_default_sized(relp_options* options, int32_t caller_size);
/* The exported symbol of rounds 1-5: assumes the LIBRARY's sizeof(relp_options) -- right only for a caller built against the header the
 * library was built with.  C and C++ callers get the sized call through the macro below without changing a line. */
int32_t relp_options_default(relp_options* options);
#define relp_options_default(options) relp_options_default_sized((options), (int32_t)sizeof(relp_options))

/* ---- host-only model: the provider without a device (usable on a machine with no GPU) ------------------------
 * `relp_model` is the `MatrixData` a provider call sequence would see; it needs no HIP device, so the host logic
 * (parser, standardisation, virtual slack columns) is testable on CPU.  relp_load_model uploads it to a handle. */
typedef struct relp_model relp_model;
int32_t relp_model_from_mps(const char* path, int32_t fixed_format, relp_model** out, char* error, int32_t error_capacity);
/* The same with `GeneralForm::presolve` (general_form/mod.rs:335-358 and general_form/presolve/ of the reference: fixed
 * variables, bound constraints, slack elimination, domain propagation) applied before `standardize()`, as the reference's
 * Netlib harness does (tests/netlib/mod.rs:58).  RELP_ERR_STATE when the presolve itself proves the problem infeasible,
 * unbounded or solves it completely (message in `error`).  The presolve computes in arbitrary precision, the host model is
 * 128-bit rationals: when a presolved value does not fit, the presolve is run again without the implied bounds that need
 * more than 126 (then 60) bits, and only if that fails as well is the LP loaded as the file states it -- never
 * RELP_ERR_OVERFLOW because of the presolve (relp_get_record_json reports the level). */
int32_t relp_model_from_mps_ex(const char* path, int32_t fixed_format, int32_t presolve, relp_model** out, char* error,
                               int32_t error_capacity);
/* The provider of a caller that builds the general form itself: `GeneralForm::new` (general_form/mod.rs:211-237) followed by
 * [`presolve` :335-463 ->] `standardize` (:325-332: split free variables, flip and shift to x >= 0 :506-587, b >= 0 :592-618,
 * minimise :623-633, rows ordered E | R | <= | >= :651-717) -> `derive_matrix_data` (:262-304).
 *   columns: CSC with ascending row indices and no explicit zeros, exact rationals value_num/value_den;
 *   row_kind[i]: 0 Equal, 1 Range (b_i - range_i <= a_i x <= b_i, range_i >= 0), 2 Less, 3 Greater
 *                (`RangedConstraintRelation`, data/linear_program/elements.rs); range_* is read for Range rows only;
 *   variable j: cost, optional lower / upper bound (`Variable`, general_form/mod.rs:140-165; shift 0, not flipped);
 *   variables are named X0, X1, ... ; relp_get_original_solution returns their values in this order. */
int32_t relp_model_from_general_form(int32_t maximize, int32_t nr_rows, int32_t nr_columns, const int64_t* column_start,
                                     const int32_t* row_index, const int64_t* value_num, const int64_t* value_den,
                                     const int32_t* row_kind, const int64_t* range_num, const int64_t* range_den,
                                     const int64_t* b_num, const int64_t* b_den, const int64_t* cost_num, const int64_t* cost_den,
                                     const uint8_t* has_lower, const int64_t* lower_num, const int64_t* lower_den,
                                     const uint8_t* has_upper, const int64_t* upper_num, const int64_t* upper_den,
                                     int64_t fixed_cost_num, int64_t fixed_cost_den, int32_t presolve, relp_model** out,
                                     char* error, int32_t error_capacity);
/* Any other `MatrixProvider` (matrix_provider/mod.rs:37-134), through the calls the reference's loops make on it: the host
 * pulls every column once (the device keeps them resident; the reference generates columns lazily per pivot).
 *   column(j): `MatrixProvider::column` -- writes at most `capacity` (row, num/den) entries, ascending rows, no zeros, and
 *              returns the number of non-zeros of the column (called again with a larger buffer if that exceeds capacity);
 *   cost_value(j): `MatrixProvider::cost_value`;  right_hand_side: `MatrixProvider::right_hand_side`, nr_rows values >= 0;
 *   pivot_element_indices: `PartialInitialBasis::pivot_element_indices` (matrix_provider/mod.rs) -- (row, column) pairs whose
 *              column is the unit vector of that row; NULL, or a count of 0, for a provider without an initial basis (phase one
 *              then starts fully artificial: `Tableau::new` over `Fully`, tableau/kind/artificial/fully.rs).
 * Variable bounds are rows of such a provider (as in examples/max_flow.rs); `implicit_bounds` has nothing to take out. */
typedef struct relp_provider {
    void* user;
    int32_t nr_rows, nr_columns;
    int32_t (*column)(void* user, int32_t j, int32_t capacity, int32_t* row, int64_t* num, int64_t* den);
    void (*cost_value)(void* user, int32_t j, int64_t* num, int64_t* den);
    void (*right_hand_side)(void* user, int64_t* num, int64_t* den);
    int32_t (*pivot_element_indices)(void* user, int32_t capacity, int32_t* rows, int32_t* columns);
} relp_provider;
int32_t relp_model_from_provider(const relp_provider* provider, relp_model** out, char* error, int32_t error_capacity);
/* Number of variables of the file and how many of them the presolve removed (0 without presolve). */
int32_t relp_model_original_variables(const relp_model* model, int32_t* nr_original, int32_t* nr_removed);
/* Graph providers (reference: examples/max_flow.rs:31-223 `Primal::new` + its MatrixProvider; examples/shortest_path.rs:20-118;
 * incidence matrix data/linear_program/network/representation.rs:24-100).  Arcs in the order the reference's column-major
 * adjacency matrix enumerates them: sorted by (tail, head), no self arcs.  `value` = capacity (max flow: arc j gets the
 * bound row V-2+j and the slack column E+j, initial pivots (V-2+j, E+j)) or length (shortest path: row of s removed,
 * b = e_t).  RELP_ERR_ARGUMENT for unsorted / duplicate / self arcs or bad terminals. */
int32_t relp_model_max_flow(int32_t nr_vertices, int32_t nr_arcs, const int32_t* tail, const int32_t* head,
                            const int64_t* capacity_num, const int64_t* capacity_den, int32_t s, int32_t t,
                            relp_model** out, char* error, int32_t error_capacity);
int32_t relp_model_shortest_path(int32_t nr_vertices, int32_t nr_arcs, const int32_t* tail, const int32_t* head,
                                 const int64_t* length_num, const int64_t* length_den, int32_t s, int32_t t,
                                 relp_model** out, char* error, int32_t error_capacity);
int32_t relp_model_free(relp_model* model);
int32_t relp_model_dimensions(const relp_model* model, int32_t* nr_rows, int32_t* nr_columns, int32_t* nr_constraints,
                              int32_t* nr_structural, int64_t* nnz, int32_t group_counts[4]);
int32_t relp_model_column(const relp_model* model, int32_t j, int32_t capacity, int32_t* count,
                          int32_t* row_index, double* value);
/* exact form of one column: values as decimal strings "num/den" joined by ';' are avoided -- int64 pairs instead;
 * returns RELP_ERR_OVERFLOW when a value does not fit int64. */
int32_t relp_model_column_exact(const relp_model* model, int32_t j, int32_t capacity, int32_t* count,
                                int32_t* row_index, int64_t* num, int64_t* den);
int32_t relp_model_cost(const relp_model* model, int32_t j, double* cost);
int32_t relp_model_right_hand_side(const relp_model* model, double* rhs);
int32_t relp_model_initial_pivots(const relp_model* model, int32_t capacity, int32_t* count, int32_t* rows, int32_t* columns);
int32_t relp_model_fixed_cost(const relp_model* model, double* fixed_cost);

/* Lifetime.  relp_create fails with RELP_ERR_DEVICE when no MI355X/HIP device is usable. */
int32_t relp_create(const relp_options* options, relp_handle** out);
int32_t relp_destroy(relp_handle* handle);
const char* relp_last_error(const relp_handle* handle);

/* ---- provider: replaces `MatrixData::new` (matrix_provider/matrix_data.rs:172-248) -------------------------
 * constraints: CSC over the constraint rows (sorted rows per column, no zeros), values num/den;
 * b: one per constraint row (>= 0); ranges: one per range row; upper_*: per variable, has_upper[j] != 0 adds a
 * virtual bound row + slack.  Row order must be Equality | Range | <= | >= (general_form/mod.rs:651-717). */
int32_t relp_load_matrix_data(relp_handle* handle,
                              int32_t nr_constraints, int32_t nr_variables,
                              const int64_t* column_start, const int32_t* row_index,
                              const int64_t* value_num, const int64_t* value_den,
                              const int64_t* b_num, const int64_t* b_den,
                              const int64_t* cost_num, const int64_t* cost_den,
                              const uint8_t* has_upper, const int64_t* upper_num, const int64_t* upper_den,
                              const int64_t* range_num, const int64_t* range_den,
                              int32_t nr_equality, int32_t nr_range, int32_t nr_upper, int32_t nr_lower,
                              int64_t fixed_cost_num, int64_t fixed_cost_den);

/* Dense provider `A x <= b, x >= 0, b >= 0` with integer data (column-major A, m x n): every row has its slack as an
 * initial pivot, i.e. the `FullInitialBasis` route of two_phase/mod.rs:80-109 (no phase one).  BASELINE config 3. */
int32_t relp_load_dense_le(relp_handle* handle, int32_t m, int32_t n, const int64_t* a_column_major,
                           const int64_t* b, const int64_t* cost);

/* Convenience for the step before the path: `parse_fixed`/`parse_free` + `TryInto<GeneralForm>` +
 * `standardize()` + `derive_matrix_data()` (tests/netlib/mod.rs:55-61, without presolve). */
int32_t relp_load_mps(relp_handle* handle, const char* path, int32_t fixed_format);
int32_t relp_load_mps_ex(relp_handle* handle, const char* path, int32_t fixed_format, int32_t presolve);
int32_t relp_load_model(relp_handle* handle, const relp_model* model);

/* `MatrixProvider::{nr_rows, nr_columns, nr_constraints, nr_variable_bounds}` (matrix_provider/mod.rs:37-134). */
int32_t relp_get_dimensions(const relp_handle* handle, int32_t* nr_rows, int32_t* nr_columns,
                            int32_t* nr_constraints, int32_t* nr_structural, int32_t* nr_artificial, int64_t* nnz);
/* `MatrixProvider::column(j)`; returns nnz through *count (capacity entries are written at most). */
int32_t relp_get_column(const relp_handle* handle, int32_t j, int32_t capacity, int32_t* count,
                        int32_t* row_index, double* value);
int32_t relp_get_cost(const relp_handle* handle, int32_t j, double* cost);            /* cost_value(j) */
int32_t relp_get_right_hand_side(const relp_handle* handle, double* rhs);             /* right_hand_side() */
int32_t relp_get_initial_pivots(const relp_handle* handle, int32_t capacity, int32_t* count,
                                int32_t* rows, int32_t* columns);                      /* pivot_element_indices() */

/* ---- `SolveRelaxation::solve_relaxation::<Carry<_, _>>()` (algorithm/mod.rs:17-36, two_phase/mod.rs:25-109) -- */
int32_t relp_solve_relaxation(relp_handle* handle, relp_result* result);
/* OptimizationResult::FiniteOptimum(x) after `reconstruct_solution` (matrix_data.rs:402-411): structural columns. */
int32_t relp_get_solution(const relp_handle* handle, double* x_structural);
/* Values of the variables of the loaded file, in file order: un-shifted, un-flipped, free variables recombined and the
 * variables a presolve removed evaluated (`GeneralForm::compute_full_solution_with_reduced_solution`,
 * general_form/mod.rs:753-771, 840-934).  `count` receives their number; RELP_ERR_ARGUMENT if capacity is too small. */
int32_t relp_get_original_solution(const relp_handle* handle, int32_t capacity, double* x, int32_t* count);
/* Exact optimal objective "num/den" incl. fixed cost (needs options.certify); returns needed length in *length. */
int32_t relp_get_objective_exact(const relp_handle* handle, char* buffer, int32_t capacity, int32_t* length);
/* `OptimizationResult::FiniteOptimum(SparseVector<RationalBig>)` (algorithm/mod.rs:43-47) in EXACT form, from the certificate's
 * solve of B x_B = b (needs options.certify and a certified finite optimum; RELP_ERR_STATE otherwise).
 *   original == 0: the sparse vector the reference's `solve_relaxation` returns, after `reconstruct_solution`
 *                  (matrix_data.rs:402-411): index = structural column of the standard form;
 *   original != 0: the `Solution` of `GeneralForm::compute_full_solution_with_reduced_solution` (general_form/mod.rs:840-934;
 *                  asserted value by value in tests/burkardt/test.rs:60-112, 143, 185): index = variable of the file, in
 *                  file order (shifts, flips, free splits and presolve removals undone in exact arithmetic).
 * Only the non-zero values are returned, ascending index: index[k] and, in `buffer`, their "num/den" texts (reduced, den > 0)
 * separated by '\n' and terminated by 0.  *count = their number, *length = bytes needed in `buffer`; with index == NULL and
 * buffer == NULL the call only reports the two sizes. */
int32_t relp_get_solution_exact(const relp_handle* handle, int32_t original, int32_t capacity, int32_t* count, int32_t* index,
                                char* buffer, int64_t buffer_capacity, int64_t* length);
/* Name of variable j of the loaded file (the `String` of `Solution::solution_values`, data/linear_program/solution.rs:15-24);
 * "Xj" for providers built without names.  *length receives the full length. */
int32_t relp_get_variable_name(const relp_handle* handle, int32_t j, char* buffer, int32_t capacity, int32_t* length);
/* One JSON object describing the last relp_solve_relaxation of this handle (the per-LP record of SURVEY.md section 5; the
 * reference has no logging at all): name, m, n, nnz, result, pivots per phase, polishes / refactorisations, wall times,
 * pivots/s, algorithmic bytes per pivot, objective (f64) and objective_exact ("num/den" when certified).  *length receives
 * the full length; at most capacity - 1 bytes are written. */
int32_t relp_get_record_json(const relp_handle* handle, char* buffer, int32_t capacity, int32_t* length);
/* ---- the loop in exact fixed-width integer arithmetic on the device ------------------------------------------------------
 * `solve_relaxation::<Carry<RationalBig, _>>` with the reference's arbitrary-precision rationals replaced by LIMBS x 64-bit
 * integers over a common denominator (exact.hip; int128 = 2 limbs, int256 = 4, ... up to 128 = 8192 bits): the SAME pivot sequence as the
 * reference (steepest edge with its tie rules, exact ratio test with Bland ties, zero-level pivots) for as long as the
 * numbers fit.  The solve starts with `first_limbs` limbs and restarts with twice as many whenever a value may not fit
 * (status 4 when `max_limbs` <= 128 is not enough).  For small LPs: one workgroup owns the whole solve.
 *   trace: (phase, entering column, pivot row, leaving column) per pivot, in the index space of relp_price;
 *   objective: the exact optimum "num/den" incl. fixed cost; basis: as relp_get_basis.
 * status: 1 optimal | 2 infeasible | 3 unbounded | 4 overflow | 5 pivot limit. */
typedef struct relp_exact_result {
    int32_t status;
    int32_t limbs;              /* limbs of the run that finished */
    int64_t pivots_phase_one;
    int64_t pivots_phase_two;
    int32_t trace_entries;
    int32_t objective_length;
    int32_t limbs_tried[6];     /* every width that was tried, in order ... */
    int32_t redundant_rows;     /* rows found redundant at the end of phase one: the reference removes them (`RemoveRows`), here
                                   each keeps its zero-level artificial basic (-1 - k in `basis`) and the phase-two row indices of
                                   `trace` count the remaining rows, as the reference's do */
    int32_t reserved;
    int64_t pivots_survived[6]; /* ... and the pivots it made before a value might not fit (or until it finished) */
} relp_exact_result;
int32_t relp_solve_exact(relp_handle* handle, int32_t first_limbs, int32_t max_limbs, int64_t max_pivots, relp_exact_result* result,
                         int32_t trace_capacity, int32_t* trace, char* objective, int32_t objective_capacity, int32_t* basis);
/* Measurement of the last relp_solve_exact on this handle, one record per width tried (no reference counterpart: the reference's
 * RationalBig has no fixed width).  `update_word_products_*` count the 64 x 64 -> 128-bit multiplications of the integer-preserving
 * update of N = D B^-1 (the dominant step): `needed` by the entries' bit bounds, `issued` by the waves (the integer-multiply roofline
 * of bench.py divides `needed` -- the algorithmic figure -- by `step_seconds[7]`; issued / needed is the waste of whole blocks).  step_seconds: x_B, reduced costs (y = c_B' N and every c~_j), arg-max, exact weights, tournament, entering
 * column, ratio test, update of N, bookkeeping, products N a_j and keys of the columns with c~_j < 0.  Returns the number of records through *count (at most `capacity` copied). */
typedef struct relp_exact_width_record {
    int32_t limbs;
    int32_t grid;                          /* workgroups of the cooperative launch */
    int64_t pivots_total_at_end;           /* pivots of the solve when this width stopped */
    double seconds;                        /* host wall time of this width */
    double step_seconds[10];
    int64_t update_word_products_needed;
    int64_t update_word_products_issued;
} relp_exact_width_record;
int32_t relp_get_exact_counters(const relp_handle* handle, relp_exact_width_record* records, int32_t capacity, int32_t* count);
/* `InverseMaintainer::basis_column_index_for_row` for all rows (provider indices; -1-k for artificial k). */
int32_t relp_get_basis(const relp_handle* handle, int32_t* basis);

/* ---- fine-grained trait parity (tests and the Rust shim) ---------------------------------------------------- */
/* `InverseMaintainer::from_basis` (inverse_maintenance/mod.rs:92-101): start phase two from the given basis
 * (provider column per row).  Also the checkpoint/resume entry. */
int32_t relp_set_basis(relp_handle* handle, const int32_t* basis_columns);
/* Phase-one start: `Tableau::<_, Partially<_>>::new` (kind/artificial/partially.rs:125-205). */
int32_t relp_begin_phase_one(relp_handle* handle);
/* `Tableau::from_artificial` hand-over (kind/non_artificial.rs:99-120, carry/mod.rs:499-525). */
int32_t relp_begin_phase_two(relp_handle* handle);
/* `BasisInverse::left_multiply_by_basis_inverse` (carry/mod.rs:98-107; lower_upper/mod.rs:180-210): FTRAN. */
int32_t relp_bi_ftran(relp_handle* handle, int32_t nnz, const int32_t* row_index, const double* value, double* out_m);
/* `BasisInverse::right_multiply_by_basis_inverse` (lower_upper/mod.rs:212-237): BTRAN. */
int32_t relp_bi_btran(relp_handle* handle, int32_t nnz, const int32_t* row_index, const double* value, double* out_m);
/* `BasisInverse::basis_inverse_row` (lower_upper/mod.rs:254-272). */
int32_t relp_bi_row(relp_handle* handle, int32_t row, double* out_m);
/* `PivotRule::select_primal_pivot_column` (strategy/pivot_rule.rs:221-241): *column = -1 when none. */
int32_t relp_price(relp_handle* handle, int32_t* column, double* relative_cost);
/* `Tableau::relative_cost(j)` for every column of the current index space (tableau/mod.rs:106-112). */
int32_t relp_relative_costs(relp_handle* handle, double* out_n);
/* Steepest-edge weights gamma_j (strategy/pivot_rule.rs:190-219); NaN for basic / artificial columns. */
int32_t relp_get_gamma(relp_handle* handle, double* out_n);
/* `Tableau::generate_column` + `select_primal_pivot_row` (tableau/mod.rs:126-130, 287-313): *row = -1 when unbounded. */
int32_t relp_ratio(relp_handle* handle, int32_t column, int32_t* row, double* out_alpha_m);
/* `Tableau::bring_into_basis(pivot_column, pivot_row, ..)` (tableau/mod.rs:139-160; `InverseMaintainer::change_basis`,
 * carry/mod.rs:561-604) with the pivot given by the caller, in the index space of relp_price / relp_ratio: the inverse, b, -pi,
 * the objective and the steepest-edge weights are updated as in any pivot of the loop.  RELP_ERR_STATE when the pivot
 * element is zero. */
int32_t relp_bring_into_basis(relp_handle* handle, int32_t column, int32_t row);
/* The last basis change: `BasisChangeComputationInfo::{pivot_row_index, pivot_column_index, leaving_column_index}`
 * (tableau/mod.rs:205-234) in the index space of relp_price, and the phase it was made in.  -1 when none was made yet. */
int32_t relp_get_last_pivot(relp_handle* handle, int32_t* phase, int32_t* column, int32_t* row, int32_t* leaving);
/* `PivotRule::after_basis_update(info, tableau)` (strategy/pivot_rule.rs:23-54; `SteepestDescentAlongObjective` :243-296): the
 * Goldfarb-Reid update of the steepest-edge weights for the last relp_bring_into_basis / relp_iterate pivot.  Inside the
 * device loop the update rides on the next pricing pass; this entry applies it now (no-op when none is pending, or for the
 * rules without state). */
int32_t relp_se_after_basis_update(relp_handle* handle);
/* `BasisInverse::should_refactor` + `invert` on demand (lower_upper/mod.rs:78-92, 249-252): polishes the resident inverse and
 * recomputes b, -pi and the objective from it; *residual_before = max |I - B^T T| found. */
int32_t relp_refactor(relp_handle* handle, double* residual_before);
/* One full iteration of phase_one.rs:134-178 / phase_two.rs:36-58, repeated `count` times on the device. */
int32_t relp_iterate(relp_handle* handle, int64_t count, int64_t* done, int32_t* stop_reason);
/* `InverseMaintainer::{b, get_objective_function_value}` (inverse_maintenance/mod.rs:240-264). */
int32_t relp_get_b(relp_handle* handle, double* out_m);
int32_t relp_get_objective(relp_handle* handle, double* objective);
int32_t relp_get_stats(const relp_handle* handle, relp_stats* stats);
int32_t relp_reset_stats(relp_handle* handle);
/* Measurement hook for bench.py: average execution time (seconds) of ONE launch of a hot-loop kernel INSIDE the real
 * pivot sequence: `repetitions` further pivots of the current phase are run, the chosen kernel of each one bracketed by
 * its own HIP start/stop event pair on the handle's stream (hipExtLaunchKernelGGL).  The solve advances.
 * which: 0 pricing pass (with the steepest-edge update) | 1 fused ftran/ratio | 2 inverse update. */
int32_t relp_profile_kernel(relp_handle* handle, int32_t which, int32_t repetitions, double* seconds_per_launch);

/* Diagnostic builds only (-DRELP_STAMPS): per-segment cycle sums of the fused kernel; zeros otherwise. */
int32_t relp_debug_stamps(relp_handle* handle, uint64_t* out64);
/* The switches and sizes of `options` for the test hooks that take no options of their own (relp_lu_factor_device), on the calling
 * thread, until the next call.  NULL restores the defaults. */
int32_t relp_debug_set_tuning(const relp_options* options);
/* Test hook of the exact simplex's update on the matrix cores (exact.hip, finish_update_entry): `count` numerators as the MFMA tiles
 * leave them -- `limbs` words each, word-major (word w of entry e at T[w * count + e]), one carry per pair of words
 * (carry[pair * count + e], added to the pair above), words[e] of them valid (a multiple of 8) -- become the entries of N: carries run
 * through, sign-extended from 64 * words[e] bits, shifted right by `shift` bits, negated where `flip`; N_out word-major, bits_out the
 * bit length of each magnitude.  limbs in {16, 32, 64, 128}.  No reference counterpart (tests only). */
int32_t relp_debug_exact_finish(int32_t device, int32_t limbs, int32_t count, const uint64_t* T, const int32_t* carry, const int32_t* words,
                                int32_t shift, int32_t flip, uint64_t* N_out, int32_t* bits_out);
/* Test hook of the word arithmetic behind a pivot's scalars (exact.hip): `count` pairs of `limbs`-word integers (entry e at a[e * limbs ..],
 * least significant word first), mode 0: out = 1 / a modulo 2^(64 limbs) for odd a (wave_inverse_odd: the inverse of D's odd part);
 * 1: out = -(a b) modulo 2^(64 limbs) (wave_mul_lo_negated: the rows' update factors); 2: out = a b.  No reference counterpart. */
int32_t relp_debug_exact_words(int32_t device, int32_t limbs, int32_t mode, int32_t count, const uint64_t* a, const uint64_t* b, uint64_t* out);
/* Test hook of the grid barrier that lets ONE cooperative launch run a whole exact solve (grid_barrier.hpp: arrivals counted per XCD,
 * one release fence per die, a watchdog that ends the launch instead of hanging the device).  mode 0, the exchange test: `rounds`
 * rounds on `grid` workgroups of 256 threads; in each, every thread stores a fresh value, the grid meets at the barrier, every thread
 * reads the values of `reads` other workgroups (rotating through all pairs; every round reads from every die) and counts what is not
 * this round's value.  mode 1: the last workgroup leaves one barrier out -- the watchdog (`limit_ticks` of 10 ns; 0 = ten seconds)
 * has to end the launch.  out8: [0] stale values seen, [1] the first as round << 32 | reader << 16 | writer (-1: none), [2] dies in use,
 * [3] ticks of 10 ns for workgroup 0's loop, [4] rounds, [5] workgroups that ran to the end, [6] the abort word (0: nobody gave up; else the
 * barrier's number), [7] workgroups found waiting then.  No reference counterpart (relp is single-threaded). */
int32_t relp_debug_grid_barrier(int32_t device, int32_t grid, int32_t rounds, int32_t reads, int32_t mode, int64_t limit_ticks, int64_t* out8);
/* Measurement hook of the exact update's MFMA tile by itself (exact.hip, mfma_update_tile with the fused epilogue): 512 workgroups x 4 waves
 * each run `tiles` tiles of `blocks` 64-byte blocks (<= limbs / 8) and `terms` terms (1: rescaled, 2: both products) on synthetic operands,
 * results shifted by `shift` bits; *seconds of the launch (tools/tile_bench.py).  limbs in {16, 32, 64, 128}.  No reference counterpart. */
int32_t relp_debug_exact_tile_bench(int32_t device, int32_t limbs, int32_t tiles, int32_t blocks, int32_t terms, int32_t shift, double* seconds);

/* ---- `BasisInverse` as an object of its own (no LP handle needed) ------------------------------------------------------
 * The reference's trait `BasisInverse` (tableau/inverse_maintenance/carry/mod.rs:69-169) and its main implementor
 * `LUDecomposition<F>` (carry/lower_upper/mod.rs:36-272): `P B Q = L U` by Markowitz pivoting
 * (lower_upper/decomposition/{mod.rs:27-143, pivoting.rs:45-81}), FTRAN / BTRAN through the sparse triangular factors,
 * Forrest-Tomlin row-eta updates (mod.rs:94-178, eta_file.rs:14-134), refactorisation after `refactor_period` updates
 * (mod.rs:249-252: 31).  The factorisation is computed on the host (f64; `pivot_threshold` is the relative pivot
 * tolerance floating point needs and the exact reference does not; `reference_ties` = 1 with threshold 0 reproduces the
 * reference's pivot choice entry for entry) and lives on the device, where the solves and updates run.
 * Vectors cross the boundary as the reference's sparse `(index, value)` pairs in, dense arrays of m doubles out.
 * Index spaces: a column is indexed by the rows of the LP (`left_multiply_by_basis_inverse` takes a provider column); a
 * row vector by the rows of the basis, i.e. by `Carry::basis_indices` positions (`right_multiply_by_basis_inverse`). */
typedef struct relp_basis_inverse relp_basis_inverse;
typedef struct relp_bi_options {
    int32_t device;
    int32_t refactor_period;   /* `should_refactor` turns true after this many `change_basis` calls (reference: 31) */
    double pivot_threshold;    /* accept a_ij as a pivot only if |a_ij| >= threshold * max_k |a_ik| (0: any non-zero) */
    int32_t reference_ties;    /* 1: pivoting.rs:60-80 exactly (minimum Markowitz count, ties by (column, row) position) */
    int32_t switches;          /* relp_switch bits (RELP_SW_BI_FACTOR_HOST, the RELP_SW_LUF_* hooks): in force during every call on the object */
} relp_bi_options;
int32_t relp_bi_options_default(relp_bi_options* options);
/* `BasisInverse::identity(m)` (carry/mod.rs:83; lower_upper/mod.rs:67-76). */
int32_t relp_bi_identity(const relp_bi_options* options, int32_t m, relp_basis_inverse** out);
/* `BasisInverse::invert(columns)` (carry/mod.rs:89-92; lower_upper/mod.rs:78-92): m columns in basis order, CSC over the
 * LP's rows.  RELP_ERR_NUMERICAL when the columns are singular. */
int32_t relp_bi_invert(const relp_bi_options* options, int32_t m, const int64_t* column_start, const int32_t* row_index,
                       const double* value, relp_basis_inverse** out);
int32_t relp_bi_free(relp_basis_inverse* bi);
const char* relp_bi_last_error(const relp_basis_inverse* bi);  /* bi == NULL: the last constructor error of this thread */
/* `BasisInverse::m()` (carry/mod.rs:168). */
int32_t relp_bi_m(const relp_basis_inverse* bi, int32_t* m);
/* `BasisInverse::left_multiply_by_basis_inverse(column)` (carry/mod.rs:123-129; lower_upper/mod.rs:180-210): B^-1 c.  The
 * object keeps the column and its spike -- the `ColumnComputationInfo` (lower_upper/mod.rs:417-432) -- for change_basis. */
int32_t relp_bi_left_multiply(relp_basis_inverse* bi, int32_t nnz, const int32_t* row_index, const double* value, double* out_m);
/* `BasisInverse::right_multiply_by_basis_inverse(row)` (carry/mod.rs:135-141; lower_upper/mod.rs:212-237): r B^-1. */
int32_t relp_bi_right_multiply(relp_basis_inverse* bi, int32_t nnz, const int32_t* index, const double* value, double* out_m);
/* `BasisInverse::basis_inverse_row(row)` (carry/mod.rs:165; lower_upper/mod.rs:254-272). */
int32_t relp_bi_basis_inverse_row(relp_basis_inverse* bi, int32_t row, double* out_m);
/* `BasisInverse::generate_element(i, original_column)` (carry/mod.rs:150-157; lower_upper/mod.rs:239-247): element i of
 * B^-1 c; *is_some = 0 when it is zero (the reference returns `None`). */
int32_t relp_bi_generate_element(relp_basis_inverse* bi, int32_t i, int32_t nnz, const int32_t* row_index, const double* value,
                                 double* element, int32_t* is_some);
/* `BasisInverse::change_basis(pivot_row_index, column)` (carry/mod.rs:104-108; lower_upper/mod.rs:94-178): the column of the
 * last relp_bi_left_multiply replaces the basis column of row `pivot_row_index` (Forrest-Tomlin update; ONLY the inverse
 * changes -- b, -pi and the objective are `Carry`'s, carry/mod.rs:561-604).  RELP_ERR_STATE without a preceding
 * left_multiply; RELP_ERR_NUMERICAL when the new basis is singular or the update area is full (poll should_refactor). */
int32_t relp_bi_change_basis(relp_basis_inverse* bi, int32_t pivot_row_index);
/* `BasisInverse::should_refactor()` (carry/mod.rs:163; lower_upper/mod.rs:249-252). */
int32_t relp_bi_should_refactor(relp_basis_inverse* bi, int32_t* should);
/* `RemoveBasisPart::remove_basis_part(indices)` (carry/mod.rs:176-180; basis_inverse_rows.rs:212-229): rows `indices` and
 * the basis columns of those rows leave the basis (redundant rows found in phase one); the rest is refactorised. */
int32_t relp_bi_remove_basis_part(relp_basis_inverse* bi, int32_t count, const int32_t* indices);
/* Sizes of the resident factor: non-zeros of L and U at the last refactorisation, the longest dependency chain of each
 * triangular solve (= LDS round trips on the device), updates since. */
int32_t relp_bi_statistics(relp_basis_inverse* bi, int64_t* nnz_lower, int64_t* nnz_upper, int32_t* depth_lower,
                           int32_t* depth_upper, int32_t* nr_updates);
/* The factors in the reference's own layout, for parity tests against its known-answer tests (lower_upper/mod.rs:36-58):
 * row/column permutation (forward), `lower_triangular` by column (lower_start has m + 1 entries), `upper_triangular` by
 * column with the rotations of all updates applied to the indices (upper_start: m + 1), `upper_diagonal`, and the eta
 * files (pivot, entries) with the indices each one had when it was made.  Every array holds at most `capacity` entries
 * (RELP_ERR_ARGUMENT otherwise); NULL skips an output. */
int32_t relp_bi_get_factors(relp_basis_inverse* bi, int64_t capacity, int32_t* row_permutation, int32_t* column_permutation,
                            int64_t* lower_start, int32_t* lower_row, double* lower_value, int64_t* upper_start,
                            int32_t* upper_row, double* upper_value, double* upper_diagonal, int32_t* nr_updates,
                            int64_t* eta_start, int32_t* eta_pivot, int32_t* eta_index, double* eta_value);
/* Host only (no device needed): the factorisation step of relp_bi_invert by itself -- `LUDecomposition::rows`
 * (decomposition/mod.rs:27-143).  L and U come back by ROW of the position space (lower_start / upper_start: m + 1 entries),
 * strictly triangular parts, U's diagonal apart; depth_* = longest dependency chain of the two triangular solves. */
int32_t relp_lu_factor_host(int32_t m, const int64_t* column_start, const int32_t* row_index, const double* value,
                            double pivot_threshold, int32_t reference_ties, int64_t capacity, int32_t* row_permutation,
                            int32_t* column_permutation, int64_t* lower_start, int32_t* lower_column, double* lower_value,
                            int64_t* upper_start, int32_t* upper_column, double* upper_value, double* upper_diagonal,
                            int32_t* depth_lower, int32_t* depth_upper);
/* The same factorisation with both triangles inverted as sparse matrices -- what RELP_CARRY_LU_INVERSE uploads at a
 * refactorisation (`LUDecomposition::invert`, lower_upper/mod.rs:78-92, then L^-1 and U^-1): lower_* = strict part of L^-1 by rows
 * (unit diagonal implied), upper_* = U^-1 by rows with its diagonal, upper_diagonal = ones.  Host only, for tests. */
int32_t relp_lu_invert_host(int32_t m, const int64_t* column_start, const int32_t* row_index, const double* value,
                            double pivot_threshold, int64_t capacity, int32_t* row_permutation, int32_t* column_permutation,
                            int64_t* lower_start, int32_t* lower_column, double* lower_value, int64_t* upper_start,
                            int32_t* upper_column, double* upper_value, double* upper_diagonal);

/* The factorisation step as the DEVICE runs it (round 4: `BasisInverse::invert` is a kernel, relp_amd/csrc/lu_factor.hip --
 * parallel independent pivots by Markowitz score with the same threshold test; with `reference_ties` one pivot per round, the
 * reference's, so the factors equal relp_lu_factor_host's and the reference's known answers entry for entry).  Same outputs as
 * relp_lu_factor_host, produced on HIP device `device` and copied back; `dense_tail`: the last rows (at most 64; 0 = none) go
 * through a dense LU out of LDS; info[32]: status, nnz(L), nnz(U), rounds, rows of the dense tail, arena peak, nnz(B), ...
 * `inverted` = 1: the two triangles inverted on the device as well (relp_lu_invert_host's outputs). */
int32_t relp_lu_factor_device(int32_t device, int32_t m, const int64_t* column_start, const int32_t* row_index, const double* value,
                              double pivot_threshold, int32_t reference_ties, int32_t dense_tail, int32_t inverted, int64_t capacity,
                              int32_t* row_permutation, int32_t* column_permutation, int64_t* lower_start, int32_t* lower_column,
                              double* lower_value, int64_t* upper_start, int32_t* upper_column, double* upper_value,
                              double* upper_diagonal, int32_t* info);

/* ---- `BasisInverse` over EXACT rationals (round 6) -----------------------------------------------------------------------
 * The reference's `BasisInverse` is generic over the field: `Carry<RationalBig, BI>` is what its tests run (tests/netlib/mod.rs:62).
 * relp_bi_* above is the f64 instantiation; these are the nine trait methods (carry/mod.rs:69-169) for exact rationals, one operation
 * at a time, so that the reference's known-answer tests (lower_upper/mod.rs:536-939, decomposition/mod.rs:319-438) hold with `==`
 * and a Rust shim can offer `Carry<RationalBig, GpuExact>`.  The object holds N = D B^-1 on the device --
 * integer entries over ONE positive denominator, fixed-width two's complement words, the representation of relp_solve_exact -- and
 * `change_basis` is Edmonds' integer-preserving pivot (exact division by a truncated multiplication with 1 / D_odd).  A bound that
 * reaches the width doubles the words (up to 128) before anything is written; beyond that RELP_ERR_OVERFLOW.
 * Sparse vectors come in as (index, numerator, denominator > 0) -- the reference's `Rational64` input type (io/mps/number/parse.rs:46-65);
 * a result is m integer numerators of *words 64-bit words each (two's complement, least significant word first, entry e at
 * numerators[e * *words]) and one positive denominator of *words words, NOT reduced (the caller's rational type reduces).
 * `capacity_words` is the room per integer the caller's arrays have; relp_bix_result_words says what is needed now
 * (RELP_ERR_ARGUMENT with *words set when it is not enough).  `should_refactor` is always 0: nothing accumulates. */
typedef struct relp_basis_inverse_exact relp_basis_inverse_exact;
/* `BasisInverse::identity(m)` (carry/mod.rs:83). */
int32_t relp_bix_identity(int32_t device, int32_t m, relp_basis_inverse_exact** out);
/* `BasisInverse::invert(columns)` (carry/mod.rs:89-92; lower_upper/mod.rs:78-92): m columns in basis order, CSC over the LP's rows.
 * RELP_ERR_NUMERICAL when the columns are singular. */
int32_t relp_bix_invert(int32_t device, int32_t m, const int64_t* column_start, const int32_t* row_index, const int64_t* value_num,
                        const int64_t* value_den, relp_basis_inverse_exact** out);
int32_t relp_bix_free(relp_basis_inverse_exact* bi);
const char* relp_bix_last_error(const relp_basis_inverse_exact* bi);  /* bi == NULL: the last constructor error of this thread */
/* `BasisInverse::m()` (carry/mod.rs:168). */
int32_t relp_bix_m(const relp_basis_inverse_exact* bi, int32_t* m);
int32_t relp_bix_result_words(const relp_basis_inverse_exact* bi, int32_t* words);
/* `left_multiply_by_basis_inverse(column)` (carry/mod.rs:123-129): B^-1 c.  The object keeps the column for change_basis. */
int32_t relp_bix_left_multiply(relp_basis_inverse_exact* bi, int32_t nnz, const int32_t* row_index, const int64_t* value_num, const int64_t* value_den,
                               int32_t capacity_words, uint64_t* numerators, uint64_t* denominator, int32_t* words);
/* `right_multiply_by_basis_inverse(row)` (carry/mod.rs:135-141): r B^-1. */
int32_t relp_bix_right_multiply(relp_basis_inverse_exact* bi, int32_t nnz, const int32_t* index, const int64_t* value_num, const int64_t* value_den,
                                int32_t capacity_words, uint64_t* numerators, uint64_t* denominator, int32_t* words);
/* The same for a row vector whose entries are themselves results of earlier solves -- `Carry::change_basis` multiplies B^-1 a_q from the
 * left (carry/mod.rs:561-604: the work vector of the steepest-edge update), RationalBig values no int64 holds: entry e = integer of
 * `value_words` two's complement words at values[e * value_words], all over ONE positive denominator of `value_words` words (the
 * format results come in).  *words = value_words + the object's words + 2. */
int32_t relp_bix_right_multiply_words(relp_basis_inverse_exact* bi, int32_t nnz, const int32_t* index, int32_t value_words, const uint64_t* values,
                                      const uint64_t* value_denominator, int32_t capacity_words, uint64_t* numerators, uint64_t* denominator, int32_t* words);
/* `basis_inverse_row(row)` (carry/mod.rs:165). */
int32_t relp_bix_basis_inverse_row(relp_basis_inverse_exact* bi, int32_t row, int32_t capacity_words, uint64_t* numerators, uint64_t* denominator,
                                   int32_t* words);
/* `generate_element(i, original_column)` (carry/mod.rs:150-157): ONE numerator and the denominator; *is_some = 0 when it is zero. */
int32_t relp_bix_generate_element(relp_basis_inverse_exact* bi, int32_t i, int32_t nnz, const int32_t* row_index, const int64_t* value_num,
                                  const int64_t* value_den, int32_t capacity_words, uint64_t* numerators, uint64_t* denominator, int32_t* words,
                                  int32_t* is_some);
/* `change_basis(pivot_row_index, column)` (carry/mod.rs:104-108): the column of the last relp_bix_left_multiply replaces the basis column
 * of that row.  RELP_ERR_STATE without a preceding left_multiply; RELP_ERR_NUMERICAL when the pivot element is zero. */
int32_t relp_bix_change_basis(relp_basis_inverse_exact* bi, int32_t pivot_row_index);
/* `should_refactor()` (carry/mod.rs:163). */
int32_t relp_bix_should_refactor(relp_basis_inverse_exact* bi, int32_t* should);
/* `RemoveBasisPart::remove_basis_part(indices)` (carry/mod.rs:176-180; basis_inverse_rows.rs:212-229): rows `indices` and the basis
 * columns of those rows -- artificial unit columns basic on redundant rows -- leave; what is left is the inverse of the smaller basis
 * over the same denominator. */
int32_t relp_bix_remove_basis_part(relp_basis_inverse_exact* bi, int32_t count, const int32_t* indices);


/* ---- batches of independent LPs (BASELINE config 4) -------------------------------------------
 * The reference solves one LP per call on one thread (tests/netlib/mod.rs:47-71); independent LPs are the unit that shards.
 * A batch keeps every LP resident on every worker: `workers_per_device` host threads per listed device, each owning one
 * handle (= one HIP stream) per model.  relp_batch_run serves a queue of `n_tickets` tickets -- ticket t solves model
 * schedule[t] (so K passes over a suite are ONE queue and the caller chooses the order, e.g. longest first) -- drawn with an
 * atomic fetch-add, or through `next_ticket` when several processes (one rank per GPU) share one queue: it must return a
 * fresh ticket number per call (values outside [0, n_tickets) end the worker).  No data-path collective exists.
 *   entries[t]: result of ticket t (status -1: not served by this batch); workers[w]: tickets, pivots, busy / idle seconds of
 *   worker w (relp_batch_workers of them); *makespan_seconds: wall clock of the run. */
typedef struct relp_batch relp_batch;
typedef struct relp_batch_entry {
    int32_t status;            /* relp_status of relp_solve_relaxation, or -1 */
    int32_t model;             /* schedule[t] */
    int32_t worker, device;
    relp_result result;
    double start_seconds, end_seconds;  /* relative to the start of the run */
} relp_batch_entry;
typedef struct relp_batch_worker {
    int32_t device;
    int32_t tickets;
    int64_t pivots;
    double busy_seconds;       /* inside relp_solve_relaxation */
    double queue_seconds;      /* drawing tickets */
    double idle_seconds;       /* makespan - busy */
    double finish_seconds;     /* when this worker found the queue empty */
} relp_batch_worker;
int32_t relp_batch_create(const relp_model* const* models, int32_t n_models, const relp_options* options, const int32_t* devices,
                          int32_t n_devices, int32_t workers_per_device, relp_batch** out, char* error, int32_t error_capacity);
int32_t relp_batch_destroy(relp_batch* batch);
int32_t relp_batch_workers(const relp_batch* batch, int32_t* n_workers);
int32_t relp_batch_run(relp_batch* batch, const int32_t* schedule, int64_t n_tickets, int64_t (*next_ticket)(void* user), void* user,
                       relp_batch_entry* entries, relp_batch_worker* workers, double* makespan_seconds);
/* exact optimum "num/den" of ticket t of the last run (options.certify; RELP_ERR_STATE when there is none) */
int32_t relp_batch_get_objective_exact(const relp_batch* batch, int64_t ticket, char* buffer, int32_t capacity, int32_t* length);
/* the resident handle of (worker, model), e.g. for relp_get_record_json / relp_get_solution_exact after a run */
int32_t relp_batch_handle(const relp_batch* batch, int32_t worker, int32_t model, relp_handle** out);

/* Version / build info ("relp_amd <ver> gfx950"). */
const char* relp_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RELP_AMD_H */

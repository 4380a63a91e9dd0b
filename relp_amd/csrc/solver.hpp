// Device-resident two-phase revised simplex for one LP: host-side class (one handle = one stream on one GPU).
//
// Mirrors the reference's `Tableau<Carry<F, BI>, Kind>` + `PivotRule` state (tableau/mod.rs:25-39,
// carry/mod.rs:46-66, strategy/pivot_rule.rs:190-193) but keeps every array in HBM for the whole solve:
// the host only enqueues kernels and polls a control word once per `pivots_per_launch` pivots.
#pragma once
#include <cstdlib>
#include <cstring>
#include <cstddef>
#include <hip/hip_runtime.h>

#include <atomic>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/relp_amd.h"
#include "model.hpp"
#include "lu.hpp"

namespace relp {

// Exact primal solution of a certified basis: x_B[k] = numer[k] / denom for the provider column basis[k] (certify.hip).
struct ExactPrimal;
// What a handle keeps between its certificates (certify.hip): the second stream of the dual lifting (creating and destroying a
// stream costs 4 + 2 ms, as much as the rest of a certificate), the device buffers (returned to the handle, not to the driver) and
// the p-adic digit counts the last certificate of the loaded LP needed.
struct CertifyScratch {
    hipStream_t second = nullptr;
    struct Block {
        void* ptr;
        size_t bytes;
        bool busy;
    };
    std::vector<Block> blocks;
    int digit_hints[2] = {0, 0};  // (primal, dual); 0: unknown.  Reset when another LP is loaded.
    std::shared_ptr<const void> statics;  // the LP's integer scaling (certify.hip: CertifyStatic); reset when another LP is loaded
    void* take(size_t bytes);     // smallest free block that fits, else a new allocation
    void give_back(void* ptr);
    void release();               // frees everything (the handle's destructor; device must be current)
};
// (provider column, "num/den" reduced) for every basic provider column with a non-zero exact value, ascending column
std::vector<std::pair<int, std::string>> exact_primal_values(const ExactPrimal& primal);

// Device control block, written by single-workgroup kernels, polled by the host.
struct Ctl {
    int status;        // 0 running | 1 no entering column | 2 unbounded (ratio test empty) | 3 iteration budget used | 4 refactorise (LU carry)
    int q;             // entering column (device index space: artificials first)
    int p;             // pivot row
    int leaving;       // column that left the basis at row p
    int pending;       // 1: the steepest-edge update of the last pivot is applied by the next pricing pass
    int forced_q;      // >= 0: next iteration must enter this column ...
    int forced_p;      //       ... at this row (zero-level pivots of phase_one.rs:232-278)
    int last_selected; // FirstProfitableWithMemory (pivot_rule.rs:113-150)
    long long iters;   // pivots done in the current phase
    long long budget;  // stop when iters reaches this
    double cbar_q;
    double alpha_pq;
    double gamma_q;
    double xp;
    double minus_obj;  // carry/mod.rs `minus_objective`
    double residual;   // max |I - B Binv| written by the polish
    double scan_value;
    int scan_column;
    int nz_count;      // entries of the ordered non-zero list of alpha_q (written by K2, read by K3)
    int eta_count;     // deferred product form (dense pipeline): pivots not yet folded into the stored inverse
    int touched_count; // columns of the stored inverse that are not unit vectors any more (entries of DeviceLP::tlist)
    double flip_cost;  // implicit bounds: sum of ub_j c_j over the complemented variables (current phase's costs)
    long long bound_flips;  // iterations that moved the entering variable to its other bound without a basis change
    int k2_forced;     // multi-workgroup ratio test: the pivot row of this iteration was given by the caller
    int t_buf;         // fused pivot kernel: which of the two buffers holds the current inverse (0 outside a batch)
    int eta_version;   // deferred product form: pivots made; the kept columns of M live in eta_cols buffer (eta_version & 1)
    int eta_new;       // ... and whether the last pivot added a kept column (its row had none) -- both written by K2
    int rho_buf;       // generated columns: which half of DeviceLP::rho_bits the writers of rho_p mark (flipped by the kernel that decides a pivot)
};

constexpr int ELL_W = 8;  // padded entries per column = lanes per column in the pricing kernel

enum : int { ST_RUNNING = 0, ST_NO_ENTERING = 1, ST_UNBOUNDED = 2, ST_BUDGET = 3, ST_REFACTOR = 4, ST_REFACTOR_FAILED = 5 };  // 4: LU carry, refactorise now;
// 5: the refactorisation kernels gave up (a capacity, a row too long for the eliminating wave): the pivots behind them are no-ops, the host factorises

struct DeviceLP {
    int m = 0, n = 0, n_art = 0, ld = 0;
    // dense block: device columns [dense_first, dense_first + n_dense) also stored dense, column-major, ld = dense_ld
    int n_dense = 0, dense_first = 0, dense_ld = 0;
    double* dense_val = nullptr;
    signed char* dense_val8 = nullptr;  // the same block as signed bytes (every entry an integer in [-128, 127]); rows permuted within 1024-row chunks, dense_ld a multiple of 1024
    int dense_full = 0, dense_csc_start = 0;  // 1: every column of the dense block has m entries (rows 0 .. m-1 in order), the first of them at dense_csc_start in the CSC
    int dense_lane = 0;  // 1: dense_val8 in tiles of 16 columns x 64 rows instead (price_dense_lane_kernel), -pi / rho / w zero-padded to dense_ld
    float* dense_val32 = nullptr;  // the same block as float, when every entry is exactly representable (then dense_val is not allocated)
    double* alpha_part = nullptr;  // slices of the multi-block FTRAN, [n_slices][m]
    double* alpha_in = nullptr;    // their fixed-order sum (with the pending etas applied): what K2 reads when preselected
    // Deferred product form (eta_cap > 0, dense pipeline only).  The current inverse is  M * Binv  with
    //   M = E_k ... E_1 = I + sum_c (eta_cols[:, c] - e_{eta_rows[c]}) e_{eta_rows[c]}'        (k = ctl->eta_count <= eta_cap)
    // i.e. eta_cols[:, c] is column eta_rows[c] of M.  Binv itself is only rewritten every eta_cap pivots (one rank-k
    // update); per pivot the passes over it are read-only (FTRAN, and one BTRAN pass for rho_p and w together).
    int eta_cap = 0;
    double* eta_cols = nullptr;    // [2][eta_cap][ld]: two copies, the current one is copy ctl->eta_version & 1 (each pivot writes the other)
    int* eta_rows = nullptr;       // [eta_cap]
    int* eta_slot = nullptr;       // [m]: slot of row i in eta_rows, or -1
    double* eta_gather = nullptr;  // [eta_cap][m]: rows eta_rows[c] of Binv, gathered before the rank-k update
    double* eta_dot_part = nullptr;  // [eta_cap][ceil(m / 64)]: alpha_reduce_kernel's per-block shares of alpha' M[:, c]
    // Column j of the STORED inverse is still the unit vector e_j until a row-j pivot has been folded in (E e_j = e_j for
    // every eta of another row; the polish keeps such columns exactly).  touched[j] / tlist record the others, so that the
    // FTRAN and BTRAN passes and the rank-k update only stream columns that carry information.
    int* slack_of_row = nullptr;   // [m] dense pipeline whose sparse part is one single-entry column per row: that column (or -1); the
                                   // BTRAN pass then prices those columns itself (btran_pass_kernel) and no separate launch does
    int track_touched = 0;         // 1: touched / tlist are maintained (deferred product form, or m > 2048)
    int* touched = nullptr;        // [m] 0/1
    int* tlist = nullptr;          // [m] touched columns in the order they were folded in
    // CSC of [artificial identity columns | provider columns] (matrix_data.rs:291-329 materialised once)
    int* col_start = nullptr;
    int* row_index = nullptr;
    double* value = nullptr;
    // CSR of the same matrix (for the residual I - B Binv)
    int* row_start = nullptr;
    int* col_index = nullptr;
    double* row_value = nullptr;
    double* cost = nullptr;      // current phase (n)
    double* cost1 = nullptr;     // phase one: 1 on artificials
    double* cost2 = nullptr;     // phase two: provider costs
    double* rhs = nullptr;       // right_hand_side() (m)
    double* xB = nullptr;        // Carry::b (m)
    double* minus_pi = nullptr;  // Carry::minus_pi (m)
    int* basis = nullptr;        // Carry::basis_indices (m)
    int* pos = nullptr;          // column -> row (the Tableau's basis_columns set) (n); non-basic: -1 at 0 | -2 at its upper bound
                                 // (held complemented) | -3 fixed, both bounds coincide: never priced (implicit bounds)
    double* gamma = nullptr;     // steepest-edge weights (n)
    double* cb = nullptr;        // c_B (m): costs of the basic columns, written by cb_kernel for the -pi refresh
    int* cb_idx = nullptr;       // (m + 1) ordered non-zero positions of c_B; [m] = their number
    double* Binv = nullptr;      // explicit basis inverse, COLUMN-major: Binv(i, j) at [j*ld + i]
    double* Binv2 = nullptr;     // second buffer for the polish
    double* R = nullptr;         // residual I - B Binv
    double* alpha = nullptr;     // B^-1 a_q (m)
    double* rho = nullptr;       // row p of the NEW inverse (m)
    double* w = nullptr;         // w = alpha_q' Binv_old (m)
    int* nz_index = nullptr;     // rows with alpha_i != 0 (plus p), ascending (m)
    double* nz_alpha = nullptr;  // their alpha values (m)
    double* cand_key = nullptr;  // per pricing block
    int* cand_j = nullptr;
    double* cand_cbar = nullptr;
    int* cand_rows = nullptr;    // first ELL_W entries of each block's best column (so K2 needs no dependent fetch)
    double* cand_vals = nullptr;
    int* cand_len = nullptr;     // nnz of that column (> ELL_W: the rest comes from the CSC)
    // padded copy of the first ELL_W entries of every column (value 0 padding): no col_start dependency in K1
    int ell_w = ELL_W;           // padded width in use: 2 when no column has more than two entries and m is large, else ELL_W
    int* ell_rows = nullptr;
    double* prw = nullptr;       // ell_w == 2, columns with values: (-pi_r, rho_r, w_r, 0) packed per row, kept beside the three vectors by their writers
    // ... and the same as one BIT per row, two buffers of rho_words words: the writers of rho_p set bits in buffer Ctl::rho_buf,
    // the pricing pass copies that buffer into LDS (8 KB for config 5) and clears the other one, the kernel that decides the next
    // pivot flips Ctl::rho_buf.  Bits are only ever a superset of the non-zeros (a stale bit costs one gather of an exact zero).
    unsigned* rho_bits = nullptr;
    int price_unit_pairs = 0;  // RELP_PRICE_UNIT_PAIRS: the round-2 pricing pass (two lanes per arc, byte table), for A/B and the parity test
    int rho_words = 0;  // per buffer, a multiple of 4; 0: no bit table (too many rows for LDS: the byte table is gathered instead)
    unsigned char* rho_nz = nullptr;  // generated columns without the bit table: rho_r != 0, one byte per row (rho and w are gathered only where this says so)
    double* ell_vals = nullptr;
    // generated incidence columns (ell_w == 2, every value +-1, integer costs in [-127, 127]): ell_rows carries the sign in
    // bit 31 (0x7fffffff: no entry), ell_vals is not allocated, and pricing reads the cost as a signed byte
    signed char* cost8 = nullptr;   // current phase (n)
    signed char* cost8_2 = nullptr; // phase two (phase one: zeros on every priced column)
    // Implicit upper bounds (relp_options.implicit_bounds): the `VariableBound` / `SlackBound` rows of `MatrixData`
    // (matrix_data.rs:104-112: x_j + s = u_j) are not rows of the device LP; a variable at its upper bound is held in
    // complemented form x_j = u_j - x'_j, so every non-basic variable sits at zero and pricing is unchanged up to the
    // sign of the column: pos[j] == -2 marks a complemented non-basic column, flipped[j] the complemented ones (basic too).
    double* ub = nullptr;        // [n] upper bound of each device column (+inf without one); nullptr = mode off
    double* xub = nullptr;       // [m] upper bound of the variable that is basic in row i
    int* flipped = nullptr;      // [n] 1: the column is held in complemented form
    double* rhs0 = nullptr;      // [m] right-hand side of the file (rhs holds b minus the complemented columns' u_j a_j)
    double* k2_partd = nullptr;  // multi-workgroup ratio test (m > 8192): per-workgroup partial sums / minima / candidate keys
    int* k2_parti = nullptr;     //   ... candidate rows, their basic columns, non-zero counts and list offsets
    double* scratch = nullptr;   // m or n doubles for the fine-grained ops
    Ctl* ctl = nullptr;
    // fused pivot kernel (pivot_fused_kernel, small LPs): x_B, basis and control block exist twice; pivot k of a batch reads copy
    // k & 1 and writes the other (kernels.hip, K23)
    struct State {
        Ctl* ctl = nullptr;
        double* xB = nullptr;
        int* basis = nullptr;
    };
    State state[2];
    unsigned long long* dbg = nullptr;  // diagnostic builds only (-DRELP_STAMPS): per-segment cycle sums of K2
};

#ifdef __HIPCC__
// Generated incidence columns: row j of the new rho_p for the pricing pass -- its bit in this pivot's half of rho_bits (set only:
// the pricing pass clears the other half), its byte in rho_nz where that table is kept.
__device__ __forceinline__ void mark_rho_row(const DeviceLP& lp, int rho_buf, int j, double r) {
    if (lp.rho_nz) lp.rho_nz[j] = r != 0.0;
    if (r != 0.0 && lp.rho_words) atomicOr(lp.rho_bits + (size_t)rho_buf * lp.rho_words + (j >> 5), 1u << (j & 31));
}
#endif

// The calling thread's A/B switches and sizes (relp_options.switches and the fields beside it): set by the C ABI from the handle's
// options for the duration of a call (capi.cpp `guarded`), read by the launch helpers that have no handle in reach (lu_factor.hip,
// lu_device_tasks.hip, certify.hip).  Round 5: these were getenv calls; nothing that changes a kernel or a result reads the
// environment any more.
struct Tuning {
    unsigned switches = 0;
    int certify_threads = 0, luf_dense_tail = 0, luf_slack = 0, luf_lds = 0, luf_lds_arena = 0, luf_arena_cap = 0;
    bool has(unsigned bit) const { return (switches & bit) != 0; }
};
inline Tuning& thread_tuning() {
    static thread_local Tuning tuning;
    return tuning;
}
inline Tuning tuning_of(const relp_options& o) {
    Tuning t;
    t.switches = o.switches;
    t.certify_threads = o.certify_threads;
    t.luf_dense_tail = o.luf_dense_tail;
    t.luf_slack = o.luf_slack;
    t.luf_lds = o.luf_lds;
    t.luf_lds_arena = o.luf_lds_arena;
    t.luf_arena_cap = o.luf_arena_cap;
    return t;
}
// The sizes relp_options has had: round 4 (through lu_refactor), round 5 / 6 (through carry_weights_min).  A struct of any other size
// was not written by a header of this library.
inline bool known_options_size(int32_t size) {
    const int32_t round4_size = (int32_t)(offsetof(relp_options, lu_refactor) + sizeof(int32_t));
    const int32_t round5_size = (int32_t)(offsetof(relp_options, carry_weights_min) + sizeof(double));
    static_assert(offsetof(relp_options, carry_weights_min) + sizeof(double) == sizeof(relp_options), "a field was appended: give its round a size here");
    return size == round4_size || size == round5_size;
}
// The caller's options into the library's struct: as many bytes as the caller's header had (relp_options.struct_size, one of the known
// sizes), the defaults for the rest.  RELP_ERR_ARGUMENT when the struct was not initialised by relp_options_default(_sized) or comes
// from a newer header.  The tuning fields are clamped to what their comments promise.
inline int32_t adopt_options(const relp_options* options, relp_options* out) {
    relp_options_default_sized(out, (int32_t)sizeof(relp_options));
    if (!options) return RELP_OK;
    if (!known_options_size(options->struct_size)) return RELP_ERR_ARGUMENT;
    std::memcpy(out, options, (size_t)options->struct_size);
    out->struct_size = (int32_t)sizeof(relp_options);
    out->ftran_slices = std::max(0, std::min(64, out->ftran_slices));
    out->price_lds_max = std::max(0, std::min(160 * 1024, out->price_lds_max));  // (a CU of gfx950 has 160 KB of LDS)
    return RELP_OK;
}
struct TuningScope {  // the thread's tuning for the lifetime of the object
    Tuning saved;
    explicit TuningScope(const Tuning& t) : saved(thread_tuning()) { thread_tuning() = t; }
    ~TuningScope() { thread_tuning() = saved; }
};

// One run of the exact fixed-width simplex kernel (exact.hip) at one width: what `relp_get_exact_counters` reports.
constexpr int EX_PROF_WORDS = 40;
struct ExactWidthRecord {
    int limbs = 0, grid = 0;
    long long pivots_total_at_end = 0;            // pivots of the solve so far when this width stopped (overflow) or finished
    double seconds = 0.0;                         // host wall time of this width (upload or widening + kernel)
    double step_seconds[10] = {0};                // the leader's time per step of the loop (x_B, pass B, arg-max, exact weights, tournament, alpha, ratio, update, bookkeeping, pass A)
    long long update_products_needed = 0;         // 64 x 64 -> 128-bit word products of the update of N: what the entries' bit bounds ask for ...
    long long update_products_issued = 0;         // ... and what the waves execute (every lane of a wave runs the longest count among its entries)
};

class Solver {
public:
    explicit Solver(const relp_options& options);
    ~Solver();

    void load(StandardForm&& form);
    bool loaded() const { return loaded_; }
    bool ratio_textbook() const { return ratio_textbook_; }

    void solve(relp_result* result);
    void begin_phase_one();
    void begin_phase_two();
    void set_basis(const int* basis_columns);
    long long iterate(long long count, int* stop_reason);

    // fine-grained ops (trait parity)
    void ftran(int nnz, const int* rows, const double* values, double* out);
    void btran(int nnz, const int* rows, const double* values, double* out);
    void inverse_row(int row, double* out);
    void price(int* column, double* cbar);
    void relative_costs(double* out);
    void get_gamma(double* out);
    void ratio(int column, int* row, double* alpha_out);
    void bring_into_basis(int column, int row);
    void after_basis_update();
    void solve_exact(int first_limbs, int max_limbs, long long max_pivots, int trace_capacity, int* status, int* limbs, long long* p1,
                     long long* p2, std::vector<int>* trace, std::string* objective, std::vector<int>* basis,
                     std::vector<std::pair<int, long long>>* survived, int* redundant_rows = nullptr);
    const std::vector<ExactWidthRecord>& exact_records() const { return exact_records_; }  // of the last solve_exact, one per width tried
    void last_pivot(int* phase, int* column, int* row, int* leaving);
    double refactor();
    void get_b(double* out);
    double objective();
    void get_basis(int* out);
    void get_solution(double* x) const;
    double profile_kernel(int which, int repetitions);
    void debug_stamps(unsigned long long* out64);

    const StandardForm& form() const { return form_; }
    const DeviceLP& device() const { return d_; }
    const relp_stats& stats() const { return stats_; }
    void reset_stats();
    int n_art() const { return d_.n_art; }
    bool refactors_on_device() const { return device_refactor_; }
    long long device_refactor_fallbacks() const { return device_refactor_failures_; }
    bool refactors_asynchronously() const { return async_refactor_; }
    long long async_refactors() const { return async_refactors_; }
    long long async_refactors_abandoned() const { return async_abandoned_; }
    double async_worst_residual() const { return async_worst_residual_; }
    std::string exact_objective;  // filled by certify()
    // exact x_B of the certified optimal basis (certify.hip keeps the big integers; strings are made on demand)
    std::shared_ptr<const ExactPrimal> exact_primal;
    std::string last_error;
    relp_result last_result{};

private:
    void upload();
    void free_device();
    void set_phase(int phase);
    void launch_pivots(int count, bool forced = false);
    void enqueue_price_fused(int parity);
    void enqueue_pivot_fused(int parity);  // forced: the caller set forced_q / forced_p (three-kernel pivot)
    bool fused_ = false;          // ratio test + inverse update in one launch (pivot_fused_kernel: m <= 2048, explicit carry, no implicit bounds)
    void enqueue_price(int skip_weights, bool first_of_batch = true);
    bool slack_in_btran_ = false; // the slack columns of the dense pipeline are priced by the BTRAN pass of the previous pivot
    void enqueue_ftran_ratio(int mode);
    void enqueue_update();
    void enqueue_consolidate();
    void build_graph(int count);
    void polish(bool refresh_vectors, bool force = false);
    void invert_from_scratch();
    std::vector<int> explicit_basis(const std::vector<int>& basis, const std::vector<int>& pos) const;  // implicit bounds -> basis of the reference's formulation
    void resolve_fixed_columns(std::vector<int>& pos);
    std::vector<char> zero_width_;  // implicit bounds: device columns with upper bound 0 (fixed variables)
    void ensure_polish_buffers();  // second inverse + residual matrix, allocated when a polish first has something to correct
    Ctl read_ctl();
    void write_ctl(const Ctl& c);
    int drive_out_artificials();
    void certify(relp_result* result);
    CertifyScratch certify_scratch_;
    // LU carry (relp_options.carry == RELP_CARRY_LU)
    void refactor_lu(bool refresh_vectors, bool settle = true);  // BasisInverse::invert of the current basis: kernels on the device (lu_factor.hip), or ...
    void refactor_lu_host(bool refresh_vectors);  // ... host Markowitz + upload (relp_options.lu_refactor; the LU + Forrest-Tomlin carry; the fallback)
    std::vector<ExactWidthRecord> exact_records_;
    bool device_refactor_ = false;
    long long device_refactor_failures_ = 0;
    void lu_identity();                      // BasisInverse::identity
    LuFactors lu_sets_[2];  // (two sets of factor arrays: the asynchronous refactorisation builds the next factors in the other one)
    int lu_cur_ = 0;
    LuFactors& lu() { return lu_sets_[lu_cur_]; }
    const LuFactors& lu() const { return lu_sets_[lu_cur_]; }
    // the refactorisation beside the pivots (relp_options.lu_refactor = RELP_REFACTOR_DEVICE_ASYNC; solver.hip: start_async_refactor)
    bool async_refactor_ = false, async_in_flight_ = false;
    hipStream_t refactor_stream_ = nullptr;
    hipEvent_t ev_snapshot_ = nullptr, ev_refactored_ = nullptr;
    long long async_iters_at_snapshot_ = 0;
    int* d_basis_snapshot_ = nullptr;
    double* d_probe_ = nullptr;          // [2 m + 2] the swap guard's probe vector, its image, the residual
    double async_worst_residual_ = 0.0;
    unsigned char* d_flipped_snapshot_ = nullptr;
    long long async_refactors_ = 0, async_abandoned_ = 0;
    void start_async_refactor(long long iters_now);
    void abandon_async_flight();
    bool finish_async_refactor(long long iters_now);  // true: the handle now runs on the new factors
    bool lu_mode_ = false;
    bool lu_inverse_ = false;  // ... in its inverse-factor form (lu.hpp: L^-1, U^-1 and product-form updates)
    bool lu_is_identity_ = true;
    int refactor_period_ = 64;
    // (a negative slack selects the reference's ratio test in the kernels that implement it: the fused kernel for m <= 8192 and the LU kernel)
    double ratio_delta() const { return ratio_textbook_ ? -1.0 : opt_.harris_delta; }
    bool ratio_textbook_ = false;  // the reference's ratio test runs (relp_options.ratio_rule resolved against the data and the kernels at upload)
    int unbounded_column_ = -1;  // provider column of the ray when the result is UNBOUNDED
    long long refactors_ = 0;
    double refactor_seconds_ = 0.0;
    std::vector<int> h_col_start_, h_row_index_;  // host copy of the device CSC (basis columns for the refactorisation)
    std::vector<double> h_value_, h_rhs_;
    std::vector<int> h_row_start_, h_col_index_;  // the same matrix by rows (kept for the crash basis)
    bool crash_basis();           // relp_options.crash: triangular crash basis as the start of phase one
    bool gamma_ready_ = false;    // the next set_phase keeps the uploaded steepest-edge weights
    int crash_rows_covered_ = 0;

    relp_options opt_;
    StandardForm form_;
    DeviceLP d_;
    bool loaded_ = false;
    bool binv_identity_ = true;
    int phase_ = 0;
    int price_blocks_ = 0;        // sparse pricing workgroups
    int dense_blocks_ = 0;        // dense pricing workgroups (candidate slots follow the sparse ones)
    int ftran_slices_ = 0;        // > 0: multi-block FTRAN pipeline (select -> partial FTRAN -> fused kernel)
    int sparse_first_ = 0;        // device columns priced by the CSC kernel: [sparse_first_, n)
    int dense_entry_bytes_ = 8;   // 4 when the dense block is held as float (exactly representable entries)
    bool bounded_ = false;        // implicit upper bounds are active for the loaded LP
    bool eta_mode_ = false;       // deferred product form of the inverse (DeviceLP::eta_cap > 0)
    size_t price_lds_ = 0;
    hipStream_t stream_ = nullptr;
    // one captured batch of pivots per phase (the phases differ in a kernel argument); both survive across solves of the
    // same LP, so a solve pays for no capture or instantiation after the first
    hipGraph_t graph_[4] = {nullptr, nullptr, nullptr, nullptr};      // [phase + 2 * set of LU factors]
    hipGraphExec_t graph_exec_[4] = {nullptr, nullptr, nullptr, nullptr};
    int graph_count_[4] = {0, 0, 0, 0};
    int graph_index() const { return (phase_ == 2 ? 1 : 0) + 2 * lu_cur_; }
    void destroy_graphs();
    hipEvent_t ev_a_ = nullptr, ev_b_ = nullptr;
    relp_stats stats_{};
    std::vector<double> h_solution_;
    std::vector<int> h_basis_;
    std::vector<int> redundant_rows_;
    long long pivots_[2] = {0, 0};
    long long polishes_ = 0;
    double max_residual_ = 0.0;
    long long since_polish_ = 0;
    int polish_scale_ = 1;        // multiplier of polish_period, adapted to the drift measured at each polish

    friend struct SolverAccess;
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) acts on the CURRENT device's copy of a kernel, and a batch (relp_batch_create) may
// hold handles on several devices, created and driven by worker threads: one configuration per device, race-free.
struct PerDeviceOnce {
    std::atomic<unsigned long long> done{0};
    std::mutex mutex;
    template <class F>
    void run(F&& configure) {
        int device = 0;
        if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= 64) {  // (no bit to remember it by: configure every time)
            std::lock_guard<std::mutex> lock(mutex);
            configure();
            return;
        }
        const unsigned long long bit = 1ull << device;
        if (done.load(std::memory_order_acquire) & bit) return;
        std::lock_guard<std::mutex> lock(mutex);
        if (done.load(std::memory_order_relaxed) & bit) return;
        configure();
        done.fetch_or(bit, std::memory_order_release);
    }
};

struct DeviceError : std::runtime_error {
    explicit DeviceError(const std::string& what) : std::runtime_error(what) {}
};

#define RELP_HIP(call)                                                                                   \
    do {                                                                                                 \
        hipError_t err__ = (call);                                                                       \
        if (err__ != hipSuccess)                                                                         \
            throw ::relp::DeviceError(std::string(#call) + ": " + hipGetErrorString(err__));             \
    } while (0)

}  // namespace relp

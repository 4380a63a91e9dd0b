// Device-resident LU basis factorisation with Forrest-Tomlin updates (the reference's `LUDecomposition`,
// tableau/inverse_maintenance/carry/lower_upper/mod.rs:36-58) -- data layout and host handle.  Kernels: lu.hip.
//
// Position space: index k in [0, m) is the pivot position at refactorisation time (`P B Q = L U`: rowpos = P.forward,
// colpos = Q.forward, decomposition/mod.rs:129-133).  It is STATIC between refactorisations.  Where the reference rotates
// rows and columns physically after every update (`RotateToBack`, permutation/rotate_to_back.rs:15-122) the device keeps U
// in BORDERED form.  A Forrest-Tomlin update moves its position to the end of the logical order, so after k updates
//
//            base positions   slots 0..k-1            base positions keep their relative order (= position order);
//   U  =  [      U_bb             S       ]           slot j = the position replaced by update j (`trail_pos[j]`,
//         [       0               T       ]           `slot_of[position]`); T is k x k upper triangular BY SLOT INDEX.
//
//   U_bb  the refactorised U, both orientations, NEVER rewritten: the row and the column of a replaced position are masked
//         instead -- its row is no task of the solves any more, and the solves hold its component at zero while the base
//         block is solved, so that the stale entries of its column multiply nothing;
//   S     the spikes' entries in base rows: by column (arena, for BTRAN's dot products) and by row (`app_*`: at most one
//         entry per update and row, so a fixed stride; for FTRAN);
//   T     dense, k <= 64: the part every spike chains through -- solved by ONE wave out of LDS, not by m rows waiting on
//         each other.  A position replaced twice leaves a dead slot (zero row and column, trail_pos = -1).
// The logical index of the reference (its rotated index) is: base positions in position order, then the live slots.
// L never changes between refactorisations and is held in both orientations (rows for FTRAN, columns for BTRAN, each a
// gather).  Row etas (eta_file.rs:14-18) live in one arena: eta j = pivot position eta_pivot[j], entries
// [eta_start[j], eta_start[j+1]) with POSITIONS as indices.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "lu_factor.hpp"
#include "lu_host.hpp"

namespace relp {

enum : int { LU_N_UPDATES = 0, LU_S_TOP = 1, LU_ETA_TOP = 2, LU_FLAGS = 3, LU_PF_COUNT = 4, LU_LOG_ON = 5, LU_LOG_COUNT = 6, LU_STATE_WORDS = 16 };
constexpr int LU_LOG_CAPACITY = 40;  // pivots whose etas are logged while the next factors are built on another stream (round 5)
enum : int { LU_FLAG_UNSTABLE = 1, LU_FLAG_OVERFLOW = 2 };
constexpr int LU_MAX_SLOTS = 64;  // one wave solves T

// A triangular factor in one orientation as a TASK LIST for the level-synchronous solve (lu.hip: lu_solve_tasks).  The rows
// (columns) of the factor are sorted by dependency level.  A row without entries needs no solve (`z_pos`, `z_dinv`).  Every
// other row is packed into G = 1, 2, 4, ... 64 consecutive SLOTS of at most LU_TE entries (G = the power of two that gives each
// slot at most LU_TE entries; aligned to G, so a row never straddles a wave).  The slots are cut into CHUNKS of whole levels
// with at most LU_ROUNDS * 1024 slots (all of a Netlib-sized factor is one chunk); inside a chunk slot k belongs to thread
// k mod 1024, round k / 1024, and a thread holds ALL its slots of the chunk in registers before the first level starts (ELL
// layout: entry e of slot s at [e * stride + s]).  One barrier per level, and between two barriers ONE dependent LDS round
// trip (the operands), the multiply-adds, a DPP sum over the row's lanes and one LDS write; no global load inside a chunk.
// A row of more than 64 LU_TE entries keeps the rest in `x_idx` / `x_val` (read in the level; rare).
#ifndef RELP_LU_TE
#define RELP_LU_TE 4
#endif
constexpr int LU_TE = RELP_LU_TE;   // (a macro only so that micro-variants can be compiled side by side)
#ifndef RELP_LU_ROUNDS
#define RELP_LU_ROUNDS 1
#endif
constexpr int LU_ROUNDS = RELP_LU_ROUNDS;   // slots per thread and chunk
constexpr int LU_CHUNK_SLOTS = LU_ROUNDS * 1024;
// (global-address-space pointers: through generic pointers the compiler emits flat_load, which counts in lgkmcnt as well as vmcnt,
//  so that every LDS wait of a solve would also wait for the records still on their way from L2)
typedef const __attribute__((address_space(1))) int* lu_gptr_i32;
typedef const __attribute__((address_space(1))) double* lu_gptr_f64;
struct LuTasks {
    lu_gptr_i32 z_pos = nullptr; lu_gptr_f64 z_dinv = nullptr;  // [stride] rows without entries (padding: position 0, factor 1 ... never used past nz)
    lu_gptr_i32 s_pos = nullptr;     // [stride] position of the slot's row
    lu_gptr_i32 s_lev = nullptr;     // [stride] its level (>= 1); 0x7fffffff: no slot (all of the padding up to `stride`)
    lu_gptr_i32 s_flags = nullptr;   // [stride] log2(G) | (last lane of the group: writes the component) << 8 | (row has extra entries) << 9
    lu_gptr_f64 s_dinv = nullptr;    // [stride] 1 / diagonal of the row (U), 1 (L)
    lu_gptr_i32 s_col = nullptr; lu_gptr_f64 s_val = nullptr;   // [LU_TE][stride]; padding: position 0, value 0
    lu_gptr_i32 s_xstart = nullptr; lu_gptr_i32 s_xn = nullptr;  // [stride] extra entries of the row (G = 64 only): range in x_idx / x_val
    lu_gptr_i32 x_idx = nullptr; lu_gptr_f64 x_val = nullptr;
    lu_gptr_i32 chunk = nullptr;     // [chunks][8]: first slot, end slot, first level, end level, tail level (chunk 0 starts at slot 0).
                                     // From the tail level on every slot of the chunk sits in its LAST wave, which then runs on alone without barriers
    lu_gptr_i32 counts = nullptr;    // [LU_CNT_WORDS]: device-resident, so that a captured graph survives a refactorisation
    // The inverse-factor form reads its slots in a COMPACT record instead (a product streams every entry of the factor into one CU,
    // whose L1 moves 64 bytes per cycle: bytes are what a product costs).  44 bytes per slot instead of 56 in ten arrays:
    //   c_hdr   position of the row (16 bits) | log2 G << 16 | writes << 19 | has extra entries << 20 | wave summary << 21
    //           (bits 21-26: some row of the slot's wave spans more than 2^j lanes; bit 27: some row of it has extra entries)
    //   c_col   the four operand positions, 16 bits each;   c_val   the four values, one 32-byte piece per slot
    const __attribute__((address_space(1))) unsigned int* c_hdr = nullptr;
    const __attribute__((address_space(1))) unsigned long long* c_col = nullptr;
    lu_gptr_f64 c_val = nullptr;
    lu_gptr_i32 c_zpos = nullptr;    // rows without entries (the same list as z_pos, inside the one uploaded region)
};
enum : int { LU_CNT_Z = 0, LU_CNT_SLOTS = 1, LU_CNT_LEVELS = 2, LU_CNT_CHUNKS = 3, LU_CNT_C0_END = 4, LU_CNT_C0_L0 = 5, LU_CNT_C0_L1 = 6, LU_CNT_C0_TAIL = 7, LU_CNT_WORDS = 8 };
constexpr int LU_MAX_CHUNKS = 256;

struct DeviceLU {
    int m = 0;
    int max_updates = 0;  // slots available (<= LU_MAX_SLOTS); the refactorisation period never exceeds it
    int ldt = 0;          // leading dimension of T (max_updates + 1: odd, so that a column is conflict-free in LDS)
    int* rowpos = nullptr;
    int* colpos = nullptr;
    int* l_rstart = nullptr; int* l_rcol = nullptr; double* l_rval = nullptr;  // strict L by rows   (FTRAN gather)
    int* l_cstart = nullptr; int* l_crow = nullptr; double* l_cval = nullptr;  // strict L by columns (BTRAN gather)
    int* u_rstart = nullptr; int* u_rcol = nullptr; double* u_rval = nullptr;  // U_bb by rows    (CSR, m + 1 starts; never rewritten)
    int* u_cstart = nullptr; int* u_crow = nullptr; double* u_cval = nullptr;  // U_bb by columns
    int* app_len = nullptr; int* app_slot = nullptr; double* app_val = nullptr;                       // S by rows, stride max_updates
    int* s_cstart = nullptr; int* s_clen = nullptr; int* s_crow = nullptr; double* s_cval = nullptr;  // S by columns (arena)
    int s_capacity = 0;
    double* T = nullptr;        // [max_updates * ldt], T[a * ldt + b] = U(slot a, slot b), a < b
    int* trail_pos = nullptr;   // [max_updates] position of slot j, -1: dead
    int* slot_of = nullptr;     // [m] live slot of a position, -1: base
    double* diag = nullptr;
    int* eta_start = nullptr; int* eta_pivot = nullptr; int* eta_idx = nullptr; double* eta_val = nullptr;
    int eta_capacity = 0;
    // The etas are applied in parallel (lu.hip: lu_etas_forward / lu_etas_backward).  An entry of eta j at a position that an
    // EARLIER eta i pivots on does not go to the arena but to MF[j][i] (i the latest such eta; -1 there when i pivots on eta j's
    // own position): what chains the etas together is then a k x k unit lower-triangular matrix, solved by one wave out of
    // registers, and everything else of an eta is an independent dot product (FTRAN) or a gather by position (BTRAN).
    double* eta_mf = nullptr;     // [max_updates * ldt], MF[j * ldt + i], i < j
    int* eta_of_pos = nullptr;    // [m] latest eta that pivots on the position, -1: none
    int* eta_first = nullptr;     // [m] first such eta, -1: none
    int* eta_prev = nullptr;      // [max_updates] per eta: the previous eta with the same pivot, -1: none
    int* eapp_len = nullptr; int* eapp_eta = nullptr; double* eapp_val = nullptr;  // arena entries by POSITION, stride max_updates
    double* spike = nullptr;  // [m] position space: the FTRAN intermediate before the U solve (mod.rs:196 `spike`)
    // task lists of the four triangular solves: 0 L by rows, 1 U by rows (FTRAN), 2 U by columns, 3 L by columns (BTRAN)
    LuTasks tasks[4];         // (kernel arguments: read from the kernarg segment where they are used, no dependent round trip)
    int task_stride = 0;      // ELL stride of the slots (their capacity: a multiple of 1024, at least 1024 past the last slot)
    int* state = nullptr;     // LU_* words
    // ---- the inverse-factor form (relp_options.carry = RELP_CARRY_LU_INVERSE) -------------------------------------------------------
    // The four task lists then hold L^-1 (strict part) and U^-1 (diagonal included) instead of L and U -- every row in ONE level:
    // a solve is a sparse matrix-vector product out of place (x0 -> x1 -> x0), no level loop, no barrier between its rows.  The
    // factors are never touched between refactorisations; the updates are kept in PRODUCT FORM on top of them,
    //     B_k^-1 = M B_0^-1,   M = E_k ... E_1,   E = I - (alpha - e_p) e_p' / alpha_p   (the eta of a pivot on basis slot p),
    // with M held as the (at most max_updates) columns in which it differs from the identity (`pf_M`, column c belongs to basis
    // slot pf_slot[c]): applying M or M' is k multiply-adds per row, all rows at once -- no sequential eta loop --, and a pivot
    // folds its eta into the kept columns (m x k multiply-adds).  state[LU_PF_COUNT] = k, state[LU_N_UPDATES] = pivots since the
    // refactorisation.
    int inverse_factors = 0;
    // Round 5, the refactorisation beside the pivots: while state[LU_LOG_ON] the pivot kernel also writes the row factors
    // (alpha_s - [s == p]) / alpha_p of its eta and p to log_factor[state[LU_LOG_COUNT]] / log_p[..]; when the new factors (of the
    // basis as it was when the log began) are ready, lu_replay_kernel folds the logged etas into THEIR product form:
    // B_now^-1 = E_k ... E_1 B_snapshot^-1 whatever factors the alphas were computed with.
    double* log_factor = nullptr;  // [LU_LOG_CAPACITY][pf_ld]
    int* log_p = nullptr;          // [LU_LOG_CAPACITY]
    double* log_w = nullptr;       // [LU_LOG_CAPACITY][LU_MAX_SLOTS] scratch of the replay: row p_e of M as eta e finds it (lu_replay_plan_kernel)
    int* log_plan = nullptr;       // [4] kept columns before / after the replay, etas folded
    double* pf_M = nullptr;     // [max_updates][pf_ld]
    int pf_ld = 0;
    int* pf_slot = nullptr;     // [max_updates] basis slot of kept column c
    int* pf_col_of = nullptr;   // [m] kept column of a basis slot, -1: none
};

// Owns the device (and pinned staging) memory of one factorisation; re-used across refactorisations.
class LuFactors {
public:
    LuFactors() = default;
    ~LuFactors();
    LuFactors(const LuFactors&) = delete;
    LuFactors& operator=(const LuFactors&) = delete;
    // uploads the factors and resets the update state; stream-ordered.  Returns true when the device layout (the addresses
    // in device()) changed: it depends on capacities only, so a refactorisation normally keeps it.
    bool upload(const HostLU& f, int max_updates, hipStream_t stream, bool inverse_factors = false);
    // The refactorisation as kernels (lu_factor.hip, lu_device_tasks.hip).  prepare_device sizes every array by BOUNDS (the factors
    // do not exist on the host): `nnz_bound` = entries a basis can have.  Returns true when the device layout changed.
    bool prepare_device(int m, int max_updates, bool inverse_factors, size_t nnz_bound);
    // Enqueues factorisation -> (inverse-factor form: inversion of the triangles, compact records) -> reset of the update state.
    // A failure leaves `failed_status` in *ctl (the pivots enqueued behind become no-ops) and its code in the info words.
    void refactor_device(const LuFactorSource& src, double threshold, int reference_ties, int dense_tail, Ctl* ctl, int failed_status,
                         hipStream_t stream);
    // Round 5: the log of the pivots' etas on this object's factors (see DeviceLU::log_factor) and their replay onto a fresh
    // factorisation of the basis the log began at.
    void start_log(hipStream_t stream);
    void replay_log_of(const LuFactors& old, hipStream_t stream);
    bool device_prepared() const { return device_prepared_; }
    const int* device_info() const { return scratch_.work().info; }  // LUF_* words of the last device refactorisation
    const DeviceLU& device() const { return d_; }
    size_t lds_bytes(int nrhs) const;  // dynamic LDS of the solve kernels for this m
    long long nnz_l = 0, nnz_u = 0;
    long long nnz_l_inverse = 0, nnz_u_inverse = 0;  // (inverse-factor form)
    int depth_l = 0, depth_u = 0;

private:
    void reserve(size_t device_bytes, size_t staging_bytes);
    DeviceLU d_;
    char* dev_ = nullptr;
    size_t dev_capacity_ = 0;
    char* staging_ = nullptr;
    size_t staging_capacity_ = 0;
    size_t cap_l_ = 0, cap_u_ = 0, cap_slots_ = 0;
    LuFactorScratch scratch_;
    bool device_prepared_ = false;
};

// kernels (lu.hip); all single-workgroup, stream-ordered
constexpr int LU_THREADS = 1024;
bool lu_fits_lds(int m, int max_updates, bool inverse_factors = false);  // max_updates: the update slots the kernels will be given (T is max_updates^2 doubles of LDS)
// FTRAN of a sparse column (device arrays rows / vals, original row indices): out[slot] (m doubles); the spike stays in lu.spike
void launch_lu_ftran(const DeviceLU& lu, const int* rows, const double* vals, int nnz, double* out, int keep_spike, hipStream_t s);
// FTRAN of a dense right-hand side (original row order)
void launch_lu_ftran_dense(const DeviceLU& lu, const double* rhs, double* out, hipStream_t s);
// round 5 (the refactorisation beside the pivots): a fixed probe vector, and max |B x - v| over the basis as it stands
void launch_lu_probe_fill(double* v, int m, hipStream_t s);
void launch_lu_basis_residual(const int* col_start, const int* row_index, const double* value, const int* basis, const int* flipped, const double* x, const double* v,
                              int m, double* out, hipStream_t s);
// BTRAN of a sparse / dense row vector given per basis slot; out per original row
void launch_lu_btran(const DeviceLU& lu, const int* slots, const double* vals, int nnz, double* out, hipStream_t s);
void launch_lu_btran_dense(const DeviceLU& lu, const double* in_slots, double* out, hipStream_t s);
// Forrest-Tomlin update for pivot slot p with the spike of the last FTRAN (mod.rs:94-178)
void launch_lu_update(const DeviceLU& lu, int p, hipStream_t s);
// the LU carry inside the device-resident loop (solver.hip)
struct DeviceLP;
void launch_lu_pivot(const DeviceLP& d, const DeviceLU& lu, int rule, int n_price_blocks, double tol_pivot, double harris_delta,
                     int skip_art, int mode, int refactor_period, hipStream_t s, hipEvent_t start, hipEvent_t stop);
void launch_lu_xb(const DeviceLP& d, const DeviceLU& lu, hipStream_t s);
void launch_lu_pi(const DeviceLP& d, const DeviceLU& lu, hipStream_t s);
void launch_lu_gamma(const DeviceLP& d, const DeviceLU& lu, hipStream_t s);
void launch_lu_row_scan(const DeviceLP& d, const double* rowvec, double tol, hipStream_t s);


// ---------------------------------------------------------------------------------------------------------------------
// A `BasisInverse` that exists without a loaded LP: the reference's `LUDecomposition<F>` as an object
// (carry/mod.rs:69-169 trait; lower_upper/mod.rs:60-272 impl).  One handle = one HIP stream on one device.
// ---------------------------------------------------------------------------------------------------------------------
class LuBasis {
public:
    LuBasis(int device, int m, const LuOptions& options, int refactor_period);
    ~LuBasis();
    int m() const { return m_; }
    void identity();                                                                     // BasisInverse::identity
    void invert(const long long* col_start, const int* rows, const double* vals);       // BasisInverse::invert
    void left_multiply(int nnz, const int* rows, const double* vals, double* out_m);    // FTRAN; keeps column + spike
    void right_multiply(int nnz, const int* slots, const double* vals, double* out_m);  // BTRAN
    void basis_inverse_row(int slot, double* out_m);
    bool generate_element(int i, int nnz, const int* rows, const double* vals, double* out);  // false: structural zero
    void change_basis(int pivot_row);                                                    // Forrest-Tomlin, last left_multiply
    bool should_refactor();                                                              // updates > period - 1
    void remove_basis_part(int count, const int* indices);                               // RemoveBasisPart (refactors)
    int updates();
    int flags();
    // Factors in the reference's layout (indices through the rotations so far), for parity tests: see relp_bi_get_factors.
    struct Factors {
        std::vector<int> row_permutation, column_permutation;
        std::vector<long long> l_start, u_start, eta_start;
        std::vector<int> l_row, u_row, eta_pivot, eta_index;
        std::vector<double> l_val, u_val, upper_diagonal, eta_value;
    };
    Factors factors();
    long long nnz_l() const { return lu_.nnz_l; }
    long long nnz_u() const { return lu_.nnz_u; }
    int depth_l() const { return lu_.depth_l; }
    int depth_u() const { return lu_.depth_u; }

private:
    void factor_and_upload();
    bool factor_on_device(const std::vector<int>& cs, const std::vector<int>& rows, const std::vector<double>& vals, HostLU& f);
    LuFactorScratch factor_scratch_;
    int device_, m_, period_;
    LuOptions options_;
    hipStream_t stream_ = nullptr;
    LuFactors lu_;
    std::vector<std::vector<std::pair<int, double>>> columns_;  // current basis columns, slot order (original rows)
    std::vector<std::pair<int, double>> last_column_;
    bool have_spike_ = false;
    int* d_idx_ = nullptr;
    double* d_val_ = nullptr;
    double* d_out_ = nullptr;
};

}  // namespace relp

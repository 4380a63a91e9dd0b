// `BasisInverse::invert` on the device: sparse LU factorisation `P B Q = L U` of the current basis as a kernel (lu_factor.hip).
//
// Replaces (paths relative to /root/reference/src/algorithm/two_phase/tableau/inverse_maintenance/carry/lower_upper/):
//   LUDecomposition::invert / LUDecomposition::rows     mod.rs:78-92, decomposition/mod.rs:27-143
//   Markowitz::choose_pivot                             decomposition/pivoting.rs:45-81
//   subtract_multiple_of_row_from_other_row             decomposition/mod.rs:146-210
// and, until round 4, `lu_factor` of lu_host.hpp (one host core, a basis read-back before and an upload after).
//
// The reference eliminates one pivot at a time: search all remaining entries for the minimum of (r_i - 1)(c_j - 1), swap it to
// (k, k), subtract the pivot row from the rows below.  On the device ONE workgroup owns a factorisation and a ROUND eliminates a
// whole set of pivots at once:
//   * every active row proposes its best admissible entry (Markowitz score, relative magnitude >= threshold as the host code;
//     a column singleton is always admissible);
//   * the candidates whose score is within a slack of the round's minimum compete: a candidate (i, j) loses to a better one
//     (i', j') when a_ij' != 0 or a_i'j != 0 -- the survivors are pairwise COMPATIBLE, their pivot block is diagonal, so they
//     can be eliminated in any order and all of their row operations commute (Davis & Yew's independent pivots);
//   * one wave per target row subtracts every pivot row it needs (in position order: deterministic rounding) with the target row
//     in registers -- four entries per lane, the pivot row's entries broadcast one by one, a match is one compare per lane;
//   * the active sub-matrix is rewritten compactly into the other of two arenas every round (its size falls quickly: on the
//     bases of a simplex run 60-80 % of the rows are singletons of the first two or three rounds), so every pass of the next
//     round walks live entries only.
// On 25FV47 a refactorisation is ~30 rounds plus a dense tail (the last <= 32 rows: partial pivoting out of LDS by one wave)
// where the sequential rule makes 821 steps.  With `reference_ties` a round accepts exactly ONE pivot, the reference's: minimum
// score, ties by the CURRENT (swapped) column then row position (pivoting.rs:60-80), threshold 0, no dense tail -- the factors
// are then the reference's entry for entry (its exact-factor known answers, decomposition/mod.rs:319-438, run on this kernel).
//
// Output: rowpos / colpos / diag and L (strict, by rows) and U (strict, by rows), position space, every row sorted by column --
// the layout of HostLU (lu_host.hpp) in device memory; lu_device_tasks.hip turns it into the task lists / compact records the
// solve kernels read, also on the device.  Nothing here depends on the order in which atomics land: entry orders inside rows
// are canonicalised by a rank sort, positions come from ordered scans.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace relp {

// where the basis columns come from: a CSC (of the whole LP: `basis[k]` picks the column of slot k; or of the basis itself:
// basis == nullptr) and, with implicit bounds, the complemented columns (opposite sign in the basis)
struct LuFactorSource {
    const int* col_start = nullptr;
    const int* row_index = nullptr;
    const double* value = nullptr;
    const int* basis = nullptr;     // [m] column of slot k, nullptr: column k
    const int* flipped = nullptr;   // [n] or nullptr
};

enum : int {
    LUF_STATUS = 0,      // 0 ok | LUF_ERR_*
    LUF_NNZ_L = 1,
    LUF_NNZ_U = 2,
    LUF_ROUNDS = 3,
    LUF_DENSE_ROWS = 4,  // rows factorised by the dense tail
    LUF_ARENA_PEAK = 5,
    LUF_NNZ_B = 6,
    LUF_NNZ_LI = 7,      // entries of L^-1 (strict) and U^-1 (with diagonal): written by the inversion kernels
    LUF_NNZ_UI = 8,
    LUF_LAYOUT = 9,      // scratch for the task builders
    LUF_LDS_ROUNDS = 10, // rounds made with the active sub-matrix in LDS
    LUF_STAMPS = 12,     // 11 cycle sums (units of 16 shader cycles): load | candidates | competition | conflicts | accept | U rows +
                         // targets | layout | copy + eliminate | reset | dense tail | finalisation
    LUF_INFO_WORDS = 32
};
enum : int {
    LUF_OK = 0,
    LUF_ERR_SINGULAR = 1,
    LUF_ERR_ARENA = 2,      // the active sub-matrix outgrew the work arena
    LUF_ERR_LONG_ROW = 3,   // a row of the active sub-matrix with more than 256 entries
    LUF_ERR_L_CAPACITY = 4,
    LUF_ERR_U_CAPACITY = 5,
    LUF_ERR_INVERSE_CAPACITY = 6,
    LUF_ERR_TASK_CAPACITY = 7,
    LUF_ERR_DATAFLOW = 8    // the inversion waited for a row that never arrived (a bug, reported instead of hanging)
};

constexpr int LUF_THREADS = 1024;
constexpr int LUF_ROW_SLOTS = 4;                  // register entries per lane of the eliminating wave
constexpr int LUF_MAX_ROW = 64 * LUF_ROW_SLOTS;   // longest row the elimination takes
constexpr int LUF_DENSE_MAX = 32;                 // the dense tail: one lane per row (its matrix sits in static LDS)

// Work memory of one factorisation (device pointers; LuFactorScratch owns them).
struct LuFactorWork {
    int m = 0;
    int cap_w = 0;                 // entries per arena
    int score_slack = 16;          // candidates with a score <= max(slack x s_min, s_min + slack) compete in a round (25FV47: 47 rounds at 4, 34 at 16, the same fill)
    int fixed_target_map = 0;      // diagnostic: the eliminating waves take targets by a fixed map instead of claiming them
    unsigned* a_cr[2] = {nullptr, nullptr};   // arena: row << 16 | column (basis slot) of the entry, 0xffffffff: hole
    double* a_val[2] = {nullptr, nullptr};
    int* r_start = nullptr;        // [m] first entry of the row in the current arena
    int* r_len = nullptr;          // [m]
    int* r_newstart = nullptr;     // [m]
    int* growth = nullptr;         // [m]
    int* active[2] = {nullptr, nullptr};  // [m] the rows not yet pivoted, ascending
    int* targets = nullptr;        // [m]
    int* ccount = nullptr;         // [m] active entries per column
    unsigned long long* rmax = nullptr;   // [m] bits of the largest magnitude of the row (atomic max while the basis is loaded)
    double* rmaxd = nullptr;       // [m] ... and as a double from then on
    unsigned* rowbest = nullptr;   // [m] best candidate of the row: score << 20 | magnitude rank << 16 | column
    int* best_e = nullptr;         // [m] arena index of that entry
    unsigned* colmark = nullptr;   // [m] the best candidate that wants the column: score << 16 | row
    int* kill = nullptr;           // [m]
    int* tflag = nullptr;          // [m]
    int* pivk_row = nullptr;       // [m] position the row got in THIS round, else -1
    int* pivk_col = nullptr;       // [m]
    // U rows as they are taken out (columns = basis slots until the end), L as (row, step, ratio) triples
    int* ut_start = nullptr;       // [m + 1]
    int* ut_col = nullptr;
    int* ut_row = nullptr;         // position of the entry's row
    double* ut_val = nullptr;
    int cap_u = 0;
    int* lt_row = nullptr;
    int* lt_step = nullptr;
    double* lt_val = nullptr;
    int cap_l = 0;
    int* tmp_start = nullptr;      // [m + 1] scratch of the finalisation
    int* tmp_cursor = nullptr;     // [m]
    int* tmp_idx = nullptr;        // [max(cap_l, cap_u)]
    double* tmp_val = nullptr;
    int* tmp_row = nullptr;
    // reference tie rule: the current positions of the unpivoted rows / columns (decomposition/mod.rs:224-273)
    int* rpos = nullptr; int* cpos = nullptr; int* row_at = nullptr; int* col_at = nullptr;
    int* info = nullptr;           // [LUF_INFO_WORDS]
};

// The factors, device resident (the arrays of DeviceLU: lu.hpp).
struct LuFactorOut {
    int* rowpos = nullptr;
    int* colpos = nullptr;
    double* diag = nullptr;
    int* l_start = nullptr; int* l_col = nullptr; double* l_val = nullptr;   // strict L by rows, m + 1 starts
    int* u_start = nullptr; int* u_col = nullptr; double* u_val = nullptr;   // strict U by rows
    int cap_l = 0, cap_u = 0;
};

// Work memory of the device-side inversion of the two triangles and of the record packing (lu_device_tasks.hip).
struct LuInverseWork {
    int m = 0;
    int cap = 0;                       // entries per inverse and orientation
    int* raw_col = nullptr; double* raw_val = nullptr; int raw_cap = 0;   // rows of both inverses as they are finished
    int* raw_start[2] = {nullptr, nullptr};
    int* raw_len[2] = {nullptr, nullptr};
    unsigned long long* raw_desc[2] = {nullptr, nullptr};   // (length + 1) << 32 | start of a finished row, 0: not finished (the dataflow's flags)
    double* acc = nullptr;             // [2][16][m] accumulators of the waves when they do not fit the LDS
    // canonical: 0 L^-1 by rows (strict), 1 U^-1 by rows (with diagonal), 2 U^-1 by columns, 3 L^-1 by columns; m + 1 starts each
    int* csr_start[4] = {nullptr, nullptr, nullptr, nullptr};
    int* csr_idx[4] = {nullptr, nullptr, nullptr, nullptr};
    double* csr_val[4] = {nullptr, nullptr, nullptr, nullptr};
    int* cursor = nullptr;             // [m + 1]
    int* tmp_idx = nullptr; int* tmp_col = nullptr; double* tmp_val = nullptr;   // [cap]
    int* row_rank = nullptr; int* row_xoff = nullptr; int* row_first = nullptr;  // [m]
    int static_rows = 0;               // diagnostic: rows by a fixed row -> wave map instead of being claimed
    int cap_extra_l = 0, cap_extra_u = 0;   // capacity of the extras arenas of the task lists (DeviceLU: x_idx / x_val)
    int* info = nullptr;               // the factorisation's info words (LuFactorWork::info)
};

class LuFactorScratch {
public:
    LuFactorScratch() = default;
    ~LuFactorScratch();
    LuFactorScratch(const LuFactorScratch&) = delete;
    LuFactorScratch& operator=(const LuFactorScratch&) = delete;
    // (re)allocates for m rows, a basis of at most nnz_basis entries and factors of at most cap_l / cap_u entries
    void reserve(int m, size_t nnz_basis, size_t cap_l, size_t cap_u, size_t cap_inverse = 0);
    const LuFactorWork& work() const { return w_; }
    const LuInverseWork& inverse_work() const { return iw_; }
    size_t bytes() const { return bytes_; }

private:
    LuFactorWork w_;
    LuInverseWork iw_;
    size_t cap_inv_ = 0;
    char* dev_ = nullptr;
    size_t bytes_ = 0;
    int m_ = 0;
    size_t nnz_ = 0, cap_l_ = 0, cap_u_ = 0;
};

// Enqueues the factorisation (one workgroup).  `dense_tail`: the last rows (<= LUF_DENSE_MAX) go through a dense LU out of LDS;
// 0 with `reference_ties`.
void launch_lu_factor(const LuFactorSource& src, const LuFactorWork& w, const LuFactorOut& out, double threshold, int reference_ties,
                      int dense_tail, hipStream_t stream);


// The inverse-factor carry's share of a refactorisation on the device (lu_device_tasks.hip): L^-1 and U^-1 from the factors, then
// the compact slot records of all four lists into the DeviceLU's task arrays.  `status_in`: the factorisation's info words
// (nothing runs when it failed); a failure anywhere leaves its code in info[LUF_STATUS] and, with `ctl`, `failed_status` in the
// control block so that the pivots enqueued behind it become no-ops and the host can fall back.
struct DeviceLU;
struct Ctl;
void launch_lu_invert(const LuFactorOut& factors, const LuInverseWork& iw, const int* status_in, hipStream_t stream);
void launch_lu_pack_inverse(const DeviceLU& lu, const LuInverseWork& iw, Ctl* ctl, int failed_status, hipStream_t stream);

}  // namespace relp

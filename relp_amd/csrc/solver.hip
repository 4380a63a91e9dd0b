// Host driver of the device-resident simplex (see solver.hpp).  Replaces, for this path:
//   two_phase/mod.rs:25-109        solve_relaxation (both the generic and the FullInitialBasis route)
//   phase_one.rs:123-278           phase-one loop + zero-level pivots
//   phase_two.rs:22-59             phase-two loop
//   kind/artificial/partially.rs   virtual artificial columns (index space: artificials first)
#include "solver.hpp"

#include <thread>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>

namespace relp {

// kernels.hip
void launch_price(const DeviceLP& d, int rule, int blocks, size_t lds, bool use_lds, int skip_weights, double tol,
                  int first, int last, int cand_offset, hipStream_t s);
void launch_price_dense(const DeviceLP& d, int blocks, int skip_weights, double tol, int cand_offset, hipStream_t s);
void configure_dense_lds(size_t lds);
int dense_lane_slots(int n_dense);
int dense_lane_ld(int m);
void launch_ftran_partial(const DeviceLP& d, int n_slices, int n_price_blocks, int rule, hipStream_t s);
bool fast_k2_available(const DeviceLP& d, int n_price_blocks);
void arm_launch_timer(int which, hipEvent_t start, hipEvent_t stop);
void take_launch_timer(int which, hipEvent_t* start, hipEvent_t* stop);
void configure_lds(size_t price_lds);
int price_columns_per_block(int ell_w, bool generated);
void launch_ftran_ratio(const DeviceLP& d, int rule, int n_price_blocks, double tol_pivot, double harris_delta,
                        int skip_artificial_rows, int mode, int n_alpha_slices, hipStream_t s);
void launch_update(const DeviceLP& d, hipStream_t s);
bool fused_pivot_available(const DeviceLP& d, int n_price_blocks);
void launch_pivot_fused(const DeviceLP& d, int rule, int parity, int n_price_blocks, double tol_pivot, double harris_delta,
                        int skip_artificial_rows, hipStream_t s);
void launch_begin_batch(const DeviceLP& d, long long add, hipStream_t s);
void launch_commit(const DeviceLP& d, int parity, hipStream_t s);
void launch_budget(const DeviceLP& d, long long add, hipStream_t s);
void launch_pi(const DeviceLP& d, hipStream_t s);
void launch_xb(const DeviceLP& d, hipStream_t s);
void launch_gamma_init(const DeviceLP& d, int identity, hipStream_t s);
void launch_identity(double* X, int m, int ld, hipStream_t s);
void launch_scatter(double* X, const long long* index, const double* value, long long count, hipStream_t s);
void launch_residual(const DeviceLP& d, const double* X, double* R, hipStream_t s);
void launch_gemm_polish(const double* X, const double* R, double* C, int m, int ld, const int* row_list, int n_rows, hipStream_t s);
void launch_residual_dense(const DeviceLP& d, double* Bd, const double* T, double* S, const int* row_list, int n_rows, hipStream_t s);
void launch_copy_rows(const double* src, double* dst, int m, int ld, const int* row_list, int n_rows, hipStream_t s);
bool gemm_row_lists_supported();
void launch_alpha_reduce(const DeviceLP& d, int n_slices, hipStream_t s);
int eta_max();
void configure_btran_lds(size_t lds);
void launch_eta_update(const DeviceLP& d, double tol_dual, hipStream_t s);
int btran_pass_blocks();
void launch_eta_consolidate(const DeviceLP& d, hipStream_t s);
void launch_mark_all_touched(const DeviceLP& d, hipStream_t s);
void launch_clear_refactor_status(const DeviceLP& d, hipStream_t s);
void launch_scaled_basis(const DeviceLP& d, double* T, double scale, hipStream_t s);
void launch_row_scan(const DeviceLP& d, int r, double tol, hipStream_t s);
void launch_ftran_vec(const DeviceLP& d, const int* rows, const double* vals, int nnz, double* out, hipStream_t s);
void launch_btran_vec(const DeviceLP& d, const int* rows, const double* vals, int nnz, double* out, hipStream_t s);
void launch_relative_cost(const DeviceLP& d, double* out, hipStream_t s);
// certify.hip
void certify_basis(const StandardForm& form, const std::vector<int>& basis_provider_columns, int device,
                   hipStream_t stream, std::string* objective, bool* certified, long long* repair_pivots,
                   std::string* message, int mode, int entering, std::shared_ptr<const ExactPrimal>* primal, CertifyScratch* scratch);

namespace {
double now_seconds() {
    using clock = std::chrono::steady_clock;
    return std::chrono::duration<double>(clock::now().time_since_epoch()).count();
}
template <class T>
T* dmalloc(size_t count) {
    T* p = nullptr;
    RELP_HIP(hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(count, 1) * sizeof(T)));
    return p;
}
void check_sparse(int nnz, const int* rows, const double* values, int m) {
    if (nnz < 0 || nnz > m) throw std::invalid_argument("nnz out of range");
    if (nnz > 0 && (!rows || !values)) throw std::invalid_argument("null sparse vector");
    for (int e = 0; e < nnz; ++e)
        if (rows[e] < 0 || rows[e] >= m) throw std::invalid_argument("index out of range");
}
template <class T>
void upload_vec(T* dst, const std::vector<T>& src, hipStream_t s) {
    if (!src.empty()) RELP_HIP(hipMemcpyAsync(dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, s));
}
}  // namespace

Solver::Solver(const relp_options& options) : opt_(options) {
    int count = 0;
    hipError_t err = hipGetDeviceCount(&count);
    if (err != hipSuccess || count <= 0)
        throw DeviceError("no HIP device available (relp_amd has no CPU fallback)");
    if (opt_.device < 0 || opt_.device >= count) throw DeviceError("device ordinal out of range");
    RELP_HIP(hipSetDevice(opt_.device));
    RELP_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    RELP_HIP(hipEventCreate(&ev_a_));
    RELP_HIP(hipEventCreate(&ev_b_));
}

Solver::~Solver() {
    (void)hipSetDevice(opt_.device);
    certify_scratch_.release();
    free_device();
    destroy_graphs();
    if (ev_a_) (void)hipEventDestroy(ev_a_);
    if (ev_b_) (void)hipEventDestroy(ev_b_);
    if (ev_snapshot_) (void)hipEventDestroy(ev_snapshot_);
    if (ev_refactored_) (void)hipEventDestroy(ev_refactored_);
    if (refactor_stream_) (void)hipStreamDestroy(refactor_stream_);
    if (d_basis_snapshot_) (void)hipFree(d_basis_snapshot_);
    if (d_probe_) (void)hipFree(d_probe_);
    if (d_flipped_snapshot_) (void)hipFree(d_flipped_snapshot_);
    if (stream_) (void)hipStreamDestroy(stream_);
}

void Solver::free_device() {
    void* ptrs[] = {d_.col_start, d_.row_index, d_.value, d_.row_start, d_.col_index, d_.row_value, d_.cost, d_.cost1,
                    d_.cost2, d_.rhs, d_.xB, d_.minus_pi, d_.basis, d_.pos, d_.gamma, d_.Binv, d_.Binv2, d_.R,
                    d_.alpha, d_.rho, d_.nz_index, d_.nz_alpha, d_.w, d_.cand_key, d_.cand_j, d_.cand_cbar, d_.cand_rows, d_.cand_vals, d_.cand_len, d_.ell_rows, d_.ell_vals, d_.scratch, d_.ctl, d_.dbg, d_.dense_val, d_.dense_val32, d_.dense_val8, d_.alpha_part, d_.alpha_in, d_.eta_cols, d_.eta_rows, d_.eta_slot, d_.eta_gather, d_.eta_dot_part, d_.touched, d_.tlist, d_.ub, d_.xub, d_.flipped, d_.rhs0, d_.k2_partd, d_.k2_parti, d_.prw, d_.rho_nz, d_.rho_bits, d_.cost8, d_.cost8_2, d_.cb, d_.cb_idx, d_.slack_of_row, d_.state[0].ctl, d_.state[0].xB, d_.state[0].basis, d_.state[1].ctl, d_.state[1].xB, d_.state[1].basis};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    d_ = DeviceLP{};
}

void Solver::reset_stats() { stats_ = relp_stats{}; }

void Solver::load(StandardForm&& form) {
    RELP_HIP(hipSetDevice(opt_.device));
    loaded_ = false;  // a failed upload must not leave a handle that looks loaded (its device pointers are gone)
    phase_ = 0;
    free_device();
    destroy_graphs();
    form_ = std::move(form);
    certify_scratch_.digit_hints[0] = certify_scratch_.digit_hints[1] = 0;  // (a new LP: nothing is known about its certificate)
    certify_scratch_.statics.reset();
    try {
        upload();
    } catch (...) {
        free_device();
        throw;
    }
    loaded_ = true;
}

// Materialise [artificials | provider columns] once (CSC + CSR) and upload.  The artificial columns are the virtual
// identity columns of `Partially::original_column` (kind/artificial/partially.rs:52-60); the slack columns are the
// virtual columns of `MatrixData::column` (matrix_data.rs:308-327): 12 bytes each, so materialising them costs nothing
// and makes the pricing pass one uniform CSC sweep.
void Solver::upload() {
    const MatrixData& md = form_.data;
    const bool timing = diagnostic("RELP_TIME_UPLOAD");  // diagnostic: where the one-off load time goes
    double t_last = now_seconds();
    auto tick = [&](const char* what) {
        if (!timing) return;
        const double t = now_seconds();
        fprintf(stderr, "[upload] %-28s %.3f s\n", what, t - t_last);
        t_last = t;
    };
    // Implicit upper bounds: the device LP has the constraint rows only (E | R | <= | >=) and the provider columns of the
    // first four groups (structurals, range slacks, <= slacks, >= slacks); the VariableBound / SlackBound rows and their
    // slack columns (matrix_data.rs:104-145) become upper bounds of the structurals and the range slacks.
    bounded_ = opt_.implicit_bounds != 0 && md.nr_variable_bounds() > 0;
    // relp_options first; the environment variables of rounds 1-3 still override (A/B runs without a new handle's options)
    const bool want_f64_block = opt_.dense_storage == RELP_DENSE_DOUBLE;
    const bool want_f32_block = opt_.dense_storage == RELP_DENSE_FLOAT;
    const bool product_form_off = opt_.product_form == 1;
    const int ftran_min_nnz_opt = opt_.ftran_min_nnz > 0 ? opt_.ftran_min_nnz : 1024;
    const auto sw = [&](unsigned bit) { return (opt_.switches & bit) != 0; };
    const int m = bounded_ ? md.nr_constraints() : md.nr_rows();
    const int n_p = bounded_ ? md.col_end[3] : md.nr_columns();
    if (m < 1) throw std::runtime_error("LP without rows");
    auto pivots = md.pivot_element_indices();
    std::vector<int> real_column_of_row(m, -1);
    for (auto& [row, column] : pivots)
        if (row < m && column < n_p) real_column_of_row[row] = column;
    std::vector<int> artificial_rows;
    for (int i = 0; i < m; ++i)
        if (real_column_of_row[i] < 0) artificial_rows.push_back(i);
    const int n_art = (int)artificial_rows.size();
    const int n = n_art + n_p;

    std::vector<int> col_start(n + 1, 0), row_index;
    std::vector<double> value;
    for (int k = 0; k < n_art; ++k) {
        row_index.push_back(artificial_rows[k]);
        value.push_back(1.0);
        col_start[k + 1] = (int)row_index.size();
    }
    for (int j = 0; j < n_p; ++j) {
        SparseColumn c = md.column(j);
        for (size_t e = 0; e < c.nnz(); ++e) {
            if (c.index[e] >= m) continue;  // the bound-row entry of a bounded column (implicit bounds)
            row_index.push_back(c.index[e]);
            value.push_back(c.value[e].to_double());
        }
        col_start[n_art + j + 1] = (int)row_index.size();
    }
    tick("columns -> CSC");
    const size_t nnz = row_index.size();
    std::vector<int> row_start(m + 1, 0), col_index(nnz);
    std::vector<double> row_value(nnz);
    for (size_t e = 0; e < nnz; ++e) row_start[row_index[e] + 1]++;
    for (int i = 0; i < m; ++i) row_start[i + 1] += row_start[i];
    {
        std::vector<int> fill(row_start.begin(), row_start.end() - 1);
        for (int j = 0; j < n; ++j)
            for (int e = col_start[j]; e < col_start[j + 1]; ++e) {
                int dst = fill[row_index[e]]++;
                col_index[dst] = j;
                row_value[dst] = value[e];
            }
    }
    std::vector<double> cost1(n, 0.0), cost2(n, 0.0), rhs(m);
    for (int k = 0; k < n_art; ++k) cost1[k] = 1.0;  // artificial::Cost::One (kind/artificial/partially.rs:42-50)
    for (int j = 0; j < n_p; ++j) cost2[n_art + j] = md.cost_value(j).to_double();
    auto rhs_exact = md.right_hand_side();
    for (int i = 0; i < m; ++i) rhs[i] = rhs_exact[i].to_double();

    tick("CSR, costs, rhs");
    d_.m = m;
    d_.n = n;
    d_.n_art = n_art;
    d_.ld = m;
    lu_mode_ = opt_.carry == RELP_CARRY_LU || opt_.carry == RELP_CARRY_LU_INVERSE;
    lu_inverse_ = opt_.carry == RELP_CARRY_LU_INVERSE;
    // (default: the reference's `should_refactor`, > 30 updates, for its Forrest-Tomlin form; 47 for the inverse-factor form, whose
    //  kept columns cost less per update than its refactorisation per pivot: 25FV47 63 -> 59 us per pivot, CYCLE 87 -> 81)
    refactor_period_ = std::min(opt_.refactor_period > 0 ? opt_.refactor_period : (lu_inverse_ ? 47 : 31), LU_MAX_SLOTS - 1);  // T is solved by one wave
    // `BasisInverse::invert` as kernels (lu_factor.hip, lu_device_tasks.hip: the inverse-factor form) or on one host core:
    // relp_options.lu_refactor, env RELP_REFACTOR=device|host.  AUTO is the host path today -- faster at every size measured.
    {
        int where = opt_.lu_refactor;
        device_refactor_ = lu_inverse_ && (where == RELP_REFACTOR_DEVICE || where == RELP_REFACTOR_DEVICE_ASYNC) && m <= 65535;
        async_refactor_ = device_refactor_ && where == RELP_REFACTOR_DEVICE_ASYNC;  // (only with the four-vector layout: checked where it starts)
    }
    // (a refactorisation on the device costs about twice the host's, so its period is the longest the kept columns allow: 25FV47 64.8 us
    //  per pivot at 47, 60.3 at 63; GREENBEA 154.5 -> 142.3)
    if (device_refactor_ && opt_.refactor_period <= 0) refactor_period_ = LU_MAX_SLOTS - 1;
    if (lu_inverse_ && !lu_fits_lds(m, refactor_period_ + 1, true))
        throw std::invalid_argument("the inverse-factor carry keeps its vectors in LDS (32 bytes per row, 24 beyond ~4300 rows): at most about 5800 rows (use the LU or the explicit carry beyond)");
    if (lu_mode_ && !lu_inverse_) {
        if (!lu_fits_lds(m, refactor_period_ + 1)) throw std::invalid_argument("the LU carry keeps its two solve vectors in LDS (16 bytes per row): at most about 8000 rows with this refactor period (use the explicit carry beyond)");
    }
    // dense block: the longest run of provider columns, starting at the first one, with nnz > m/2 (config 3: all
    // structural columns); steepest edge only (the dense kernel implements that rule)
    int n_dense = 0;
    if (opt_.pivot_rule == RELP_PIVOT_STEEPEST_EDGE && m >= 64)
        while (n_dense < n_p && (col_start[n_art + n_dense + 1] - col_start[n_art + n_dense]) * 2 > m) ++n_dense;
    if (n_dense < 64 || bounded_ || lu_mode_) n_dense = 0;  // (the dense pipeline belongs to the explicit inverse)
    d_.n_dense = n_dense;
    d_.dense_first = n_art;
    d_.dense_ld = (m + 3) & ~3;
    sparse_first_ = n_art + n_dense;
    price_lds_ = (size_t)3 * m * sizeof(double);
    {   // graph LPs (at most two entries per column) beyond the LDS-resident size: width-2 padded copy, 4x less padding to stream
        int longest = 0;
        for (int j = 0; j < n; ++j) longest = std::max(longest, col_start[j + 1] - col_start[j]);
        d_.ell_w = (longest <= 2 && n_dense == 0 && price_lds_ > 160 * 1024 - 1024 && !sw(RELP_SW_ELL_WIDE)) ? 2 : ELL_W;
    }
    // incidence columns (graph providers, examples/max_flow.rs:174-200): every value +-1 and small integer costs -- the
    // pricing pass then GENERATES the column from 8 bytes per arc (row | sign) instead of streaming 24 + 8 bytes of it
    bool unit = d_.ell_w == 2 && !sw(RELP_SW_NO_GENERATED_COLUMNS);
    for (size_t e = 0; unit && e < value.size(); ++e) unit = value[e] == 1.0 || value[e] == -1.0;
    for (int j = 0; unit && j < n; ++j) unit = cost2[j] == std::floor(cost2[j]) && std::fabs(cost2[j]) <= 127.0;
    const int cpb = price_columns_per_block(d_.ell_w, unit);
    price_blocks_ = std::min(d_.ell_w == 2 && !unit ? 2048 : 1024, (n - sparse_first_ + cpb - 1) / cpb);
    // Dense pipeline whose sparse columns are one single-entry column per row at most (the slack columns of config 3): the BTRAN
    // pass of a pivot prices them for the next one (btran_pass_kernel), one candidate slot per workgroup of that pass.
    std::vector<int> slack_of_row;
    {
        bool eligible = n_dense > 0 && opt_.pivot_rule == RELP_PIVOT_STEEPEST_EDGE && m % 2 == 0 && m <= 4096 &&
                        !product_form_off && !sw(RELP_SW_NO_SLACK_IN_BTRAN) && n > sparse_first_;
        if (eligible) {  // (the deferred product form needs the multi-block FTRAN: a column longer than its threshold)
            int longest = 0;
            for (int j = n_art; j < n; ++j) longest = std::max(longest, col_start[j + 1] - col_start[j]);
            eligible = longest > ftran_min_nnz_opt;
        }
        if (eligible) {
            slack_of_row.assign(m, -1);
            for (int j = sparse_first_; eligible && j < n; ++j) {
                eligible = col_start[j + 1] - col_start[j] == 1 && slack_of_row[row_index[col_start[j]]] < 0;
                if (eligible) slack_of_row[row_index[col_start[j]]] = j;
            }
        }
        if (!eligible) slack_of_row.clear();
        else price_blocks_ = btran_pass_blocks();
    }
    dense_blocks_ = n_dense > 0 ? std::min(opt_.dense_blocks > 0 ? opt_.dense_blocks : 256, (n_dense + 15) / 16) : 0;  // 16 waves per workgroup, one workgroup per CU (96 KB of LDS each)
    bool dense_bytes = n_dense > 0 && !want_f64_block && !want_f32_block;  // narrowest exact storage type
    for (int jd = 0; dense_bytes && jd < n_dense; ++jd)
        for (int e = col_start[n_art + jd]; dense_bytes && e < col_start[n_art + jd + 1]; ++e)
            dense_bytes = value[e] >= -128.0 && value[e] <= 127.0 && value[e] == std::floor(value[e]);
    {
        bool full = n_dense > 0;
        for (int jd = 0; full && jd < n_dense; ++jd) {
            full = col_start[n_art + jd + 1] - col_start[n_art + jd] == m;
            for (int e = col_start[n_art + jd], i = 0; full && i < m; ++e, ++i) full = row_index[e] == i;
        }
        d_.dense_full = full ? 1 : 0;
        d_.dense_csc_start = n_dense > 0 ? col_start[n_art] : 0;
    }
    bool dense_floats = n_dense > 0 && !dense_bytes && !want_f64_block;  // float holds every entry exactly
    for (int jd = 0; dense_floats && jd < n_dense; ++jd)
        for (int e = col_start[n_art + jd]; dense_floats && e < col_start[n_art + jd + 1]; ++e) dense_floats = (double)(float)value[e] == value[e];
    int vector_len = m;  // -pi, rho, w: zero-padded to the dense block's row count when the column-per-lane pricing reads them
    if (n_dense > 0 && dense_lane_slots(n_dense) <= 1024 && !sw(RELP_SW_NO_DENSE_LANE)) {
        // column-per-lane pricing: one workgroup and one candidate slot per group of 16 columns
        d_.dense_lane = 1;
        d_.dense_ld = dense_lane_ld(m);
        dense_blocks_ = dense_lane_slots(n_dense);
        vector_len = d_.dense_ld;
    }
    if (price_blocks_ + dense_blocks_ == 0) price_blocks_ = 1;
    price_lds_ = (size_t)3 * m * sizeof(double);
    int max_nnz = 0;
    for (int j = n_art; j < n; ++j) max_nnz = std::max(max_nnz, col_start[j + 1] - col_start[j]);
    ftran_slices_ = 0;
    // columns longer than this take the multi-block FTRAN pipeline (RELP_FTRAN_MIN_NNZ: test hook to exercise it on small LPs)
    const int ftran_min_nnz = ftran_min_nnz_opt;
    if (max_nnz > ftran_min_nnz && fast_k2_available(d_, price_blocks_ + dense_blocks_)) ftran_slices_ = opt_.ftran_slices > 0 ? opt_.ftran_slices : std::min(64, (max_nnz + 255) / 256);  // (4096 x 8192: 8 / 16 / 32 / 64 slices = 20.8k / 21.2k / 20.8k / 19.7k pivots/s)

    d_.col_start = dmalloc<int>(n + 1);
    d_.row_index = dmalloc<int>(nnz);
    d_.value = dmalloc<double>(nnz);
    d_.row_start = dmalloc<int>(m + 1);
    d_.col_index = dmalloc<int>(nnz);
    d_.row_value = dmalloc<double>(nnz);
    d_.cost = dmalloc<double>(n);
    d_.cost1 = dmalloc<double>(n);
    d_.cost2 = dmalloc<double>(n);
    d_.rhs = dmalloc<double>(m);
    d_.xB = dmalloc<double>(m);
    d_.minus_pi = dmalloc<double>(vector_len);
    RELP_HIP(hipMemsetAsync(d_.minus_pi, 0, (size_t)vector_len * sizeof(double), stream_));
    d_.basis = dmalloc<int>(m);
    d_.pos = dmalloc<int>(n);
    d_.gamma = dmalloc<double>(n);
    d_.cb = dmalloc<double>(m);
    d_.cb_idx = dmalloc<int>(m + 1);
    tick("sparse arrays");
    if (!lu_mode_) d_.Binv = dmalloc<double>((size_t)m * d_.ld);  // the LU carry has no m x m array at all
    // The second copy of the inverse and the residual matrix are only needed once a polish finds something to correct
    // (Solver::ensure_polish_buffers): at m = 65 534 each is 34 GB and about a second of hipMalloc, and the max-flow LP of
    // config 5, whose bases are unimodular, never needs them.
    tick("inverse buffers (hipMalloc)");
    d_.alpha = dmalloc<double>(m);
    d_.rho = dmalloc<double>(vector_len);
    RELP_HIP(hipMemsetAsync(d_.rho, 0, (size_t)vector_len * sizeof(double), stream_));
    d_.nz_index = dmalloc<int>(m);
    d_.nz_alpha = dmalloc<double>(m);
    d_.w = dmalloc<double>(vector_len);
    RELP_HIP(hipMemsetAsync(d_.w, 0, (size_t)vector_len * sizeof(double), stream_));
    d_.cand_key = dmalloc<double>(price_blocks_ + dense_blocks_);
    d_.cand_j = dmalloc<int>(price_blocks_ + dense_blocks_);
    d_.cand_cbar = dmalloc<double>(price_blocks_ + dense_blocks_);
    d_.cand_rows = dmalloc<int>((size_t)(price_blocks_ + dense_blocks_) * ELL_W);
    d_.cand_vals = dmalloc<double>((size_t)(price_blocks_ + dense_blocks_) * ELL_W);
    d_.cand_len = dmalloc<int>(price_blocks_ + dense_blocks_);
    RELP_HIP(hipMemsetAsync(d_.cand_rows, 0, (size_t)(price_blocks_ + dense_blocks_) * ELL_W * sizeof(int), stream_));  // width-2 pricing writes two of the ELL_W slots
    RELP_HIP(hipMemsetAsync(d_.cand_vals, 0, (size_t)(price_blocks_ + dense_blocks_) * ELL_W * sizeof(double), stream_));
    {
        const int width = d_.ell_w;
        std::vector<int> er((size_t)n * width, 0);
        std::vector<double> ev((size_t)n * width, 0.0);
        for (int j = 0; j < n; ++j)
            for (int e = col_start[j], k = 0; e < col_start[j + 1] && k < width; ++e, ++k) {
                er[(size_t)j * width + k] = row_index[e];
                ev[(size_t)j * width + k] = value[e];
            }
        if (unit) {  // (generated incidence columns: decided where the pricing grid was sized)
            for (int j = 0; j < n; ++j)
                for (int k = 0; k < width; ++k) {
                    const int len = col_start[j + 1] - col_start[j];
                    er[(size_t)j * width + k] = k < len ? (int)((unsigned)row_index[col_start[j] + k] | (value[col_start[j] + k] < 0.0 ? 0x80000000u : 0u))
                                                        : 0x7fffffff;
                }
            std::vector<signed char> c8(n);
            for (int j = 0; j < n; ++j) c8[j] = (signed char)cost2[j];
            d_.cost8 = dmalloc<signed char>(n);
            d_.cost8_2 = dmalloc<signed char>(n);
            upload_vec(d_.cost8_2, c8, stream_);
        } else {
            d_.ell_vals = dmalloc<double>(ev.size());
            upload_vec(d_.ell_vals, ev, stream_);
        }
        d_.ell_rows = dmalloc<int>(er.size());
        upload_vec(d_.ell_rows, er, stream_);
        RELP_HIP(hipStreamSynchronize(stream_));
    }
    d_.alpha_part = dmalloc<double>((size_t)std::max(1, ftran_slices_) * m);
    d_.alpha_in = dmalloc<double>(m);
    // deferred product form of the inverse: the dense pipeline (multi-block FTRAN), m even and <= 4096 (alpha_reduce_kernel, btran_pass_kernel)
    // (relp_options.product_form = 1 / RELP_ETA=0 keeps the per-pivot rank-one update: A/B measurements)
    eta_mode_ = n_dense > 0 && ftran_slices_ > 0 && m % 2 == 0 && m <= 4096 && !product_form_off;
    d_.eta_cap = eta_mode_ ? eta_max() : 0;
    slack_in_btran_ = eta_mode_ && !slack_of_row.empty();
    if (slack_in_btran_) {
        d_.slack_of_row = dmalloc<int>(m);
        upload_vec(d_.slack_of_row, slack_of_row, stream_);
        RELP_HIP(hipStreamSynchronize(stream_));
    }
    // unit columns of the inverse are tracked where skipping them pays: the dense pipeline and the larger sparse LPs
    // (below that the update kernel is latency bound and the extra indirection would cost a round trip)
    d_.track_touched = (eta_mode_ || m > 2048) && !lu_mode_ && !sw(RELP_SW_NO_TOUCHED) ? 1 : 0;
    d_.touched = dmalloc<int>(m);
    d_.tlist = dmalloc<int>(m);
    RELP_HIP(hipMemsetAsync(d_.touched, 0, m * sizeof(int), stream_));
    if (eta_mode_) {
        d_.eta_cols = dmalloc<double>((size_t)2 * d_.eta_cap * d_.ld);
        d_.eta_dot_part = dmalloc<double>((size_t)d_.eta_cap * ((m + 63) / 64));
        d_.eta_rows = dmalloc<int>(d_.eta_cap);
        d_.eta_slot = dmalloc<int>(m);
        d_.eta_gather = dmalloc<double>((size_t)d_.eta_cap * m);
        RELP_HIP(hipMemsetAsync(d_.eta_slot, 0xff, m * sizeof(int), stream_));
        configure_btran_lds((size_t)2 * ((m + 1) & ~1) * sizeof(double));
    }
    if (dense_bytes && d_.dense_lane) {
        const int groups = dense_blocks_, tiles_per_group = d_.dense_ld / 64;
        std::vector<signed char> bytes((size_t)groups * 16 * d_.dense_ld, 0);
        for (int jd = 0; jd < n_dense; ++jd)
            for (int e = col_start[n_art + jd]; e < col_start[n_art + jd + 1]; ++e) {
                const int row = row_index[e], within = row % 64;  // tile (group, row / 64): 64 lanes x 16 bytes; see price_dense_lane_kernel
                bytes[(((size_t)(jd / 16) * tiles_per_group + row / 64) * 64 + 16 * (within / 16) + jd % 16) * 16 + within % 16] = (signed char)value[e];
            }
        d_.dense_val8 = dmalloc<signed char>(bytes.size());
        upload_vec(d_.dense_val8, bytes, stream_);
        dense_entry_bytes_ = 1;
        RELP_HIP(hipStreamSynchronize(stream_));
    } else if (dense_bytes && (size_t)3 * ((m + 1023) & ~1023) * sizeof(double) > 160 * 1024 - 4096) {
        dense_bytes = false;  // (the row-permuted form keeps the padded vectors in LDS)
    }
    if (dense_floats && d_.dense_lane) {
        const int groups = dense_blocks_, tiles_per_group = d_.dense_ld / 64;
        std::vector<float> floats((size_t)groups * 16 * d_.dense_ld, 0.f);
        for (int jd = 0; jd < n_dense; ++jd)
            for (int e = col_start[n_art + jd]; e < col_start[n_art + jd + 1]; ++e) {
                // tile (group, row / 64): four pieces of 64 lanes x 4 floats; see price_dense_lane_kernel<true>
                const int row = row_index[e], within = row % 64, t = within % 16;
                floats[((((size_t)(jd / 16) * tiles_per_group + row / 64) * 4 + t / 4) * 64 + 16 * (within / 16) + jd % 16) * 4 + t % 4] = (float)value[e];
            }
        d_.dense_val32 = dmalloc<float>(floats.size());
        upload_vec(d_.dense_val32, floats, stream_);
        dense_entry_bytes_ = 4;
        RELP_HIP(hipStreamSynchronize(stream_));
    } else if (dense_bytes && d_.dense_lane) {
    } else if (d_.dense_lane) {
        const int groups = dense_blocks_, tiles_per_group = d_.dense_ld / 64;
        std::vector<double> doubles((size_t)groups * 16 * d_.dense_ld, 0.0);
        for (int jd = 0; jd < n_dense; ++jd)
            for (int e = col_start[n_art + jd]; e < col_start[n_art + jd + 1]; ++e) {
                // tile (group, row / 64): eight pieces of 64 lanes x 2 doubles; see price_dense_lane_kernel<8>
                const int row = row_index[e], within = row % 64, t = within % 16;
                doubles[((((size_t)(jd / 16) * tiles_per_group + row / 64) * 8 + t / 2) * 64 + 16 * (within / 16) + jd % 16) * 2 + t % 2] = value[e];
            }
        d_.dense_val = dmalloc<double>(doubles.size());
        upload_vec(d_.dense_val, doubles, stream_);
        dense_entry_bytes_ = 8;
        RELP_HIP(hipStreamSynchronize(stream_));
    } else if (dense_bytes) {
        d_.dense_ld = (m + 1023) & ~1023;
        std::vector<signed char> bytes((size_t)n_dense * d_.dense_ld, 0);
        for (int jd = 0; jd < n_dense; ++jd)
            for (int e = col_start[n_art + jd]; e < col_start[n_art + jd + 1]; ++e) {
                const int row = row_index[e], chunk = row / 1024, within = row % 1024;
                const int pair = within / 128, lane = (within % 128) / 2, t = 2 * pair + (within & 1);  // see price_dense_kernel
                bytes[(size_t)jd * d_.dense_ld + (size_t)chunk * 1024 + lane * 16 + t] = (signed char)value[e];
            }
        d_.dense_val8 = dmalloc<signed char>(bytes.size());
        upload_vec(d_.dense_val8, bytes, stream_);
        dense_entry_bytes_ = 1;
        RELP_HIP(hipStreamSynchronize(stream_));
        configure_dense_lds((size_t)3 * d_.dense_ld * sizeof(double));
    } else if (n_dense > 0) {
        std::vector<double> dense((size_t)n_dense * d_.dense_ld, 0.0);
        for (int jd = 0; jd < n_dense; ++jd)
            for (int e = col_start[n_art + jd]; e < col_start[n_art + jd + 1]; ++e) dense[(size_t)jd * d_.dense_ld + row_index[e]] = value[e];
        bool exact_in_float = !want_f64_block;  // relp_options.dense_storage = RELP_DENSE_DOUBLE keeps the f64 block
        for (size_t k = 0; exact_in_float && k < dense.size(); ++k) exact_in_float = (double)(float)dense[k] == dense[k];
        if (exact_in_float) {
            std::vector<float> dense32(dense.begin(), dense.end());
            d_.dense_val32 = dmalloc<float>(dense32.size());
            upload_vec(d_.dense_val32, dense32, stream_);
        } else {
            d_.dense_val = dmalloc<double>(dense.size());
            upload_vec(d_.dense_val, dense, stream_);
        }
        dense_entry_bytes_ = exact_in_float ? 4 : 8;
        RELP_HIP(hipStreamSynchronize(stream_));
        configure_dense_lds((size_t)3 * d_.dense_ld * sizeof(double));
    }
    d_.rhs0 = dmalloc<double>(m);
    if (bounded_) {
        std::vector<double> ub(n, std::numeric_limits<double>::infinity());
        for (int j = 0; j < md.nr_normal_variables(); ++j)
            if (md.variables[j].has_upper) ub[n_art + j] = md.variables[j].upper.to_double();
        for (int k = 0; k < md.nr_range; ++k) ub[n_art + md.col_end[0] + k] = md.ranges[k].to_double();
        zero_width_.assign(n, 0);
        for (int j = 0; j < n; ++j) zero_width_[j] = ub[j] == 0.0 ? 1 : 0;
        d_.ub = dmalloc<double>(n);
        d_.xub = dmalloc<double>(m);
        d_.flipped = dmalloc<int>(n);
        upload_vec(d_.ub, ub, stream_);
        RELP_HIP(hipStreamSynchronize(stream_));
    }
    if (!fast_k2_available(d_, price_blocks_ + dense_blocks_) && !sw(RELP_SW_K2_SINGLE)) {  // m > 8192: multi-workgroup ratio test
        d_.k2_partd = dmalloc<double>((size_t)8 * ((m + 1023) / 1024));
        d_.k2_parti = dmalloc<int>((size_t)4 * ((m + 1023) / 1024));
    }
    // `Tableau::select_primal_pivot_row` (tableau/mod.rs:287-313): which ratio test runs.  The reference's rule is implemented by the
    // register-resident ratio test (m <= 8192), the fused pivot kernel and the LU pivot kernel; the multi-workgroup test beyond 8192 rows
    // and the one-workgroup fallback implement the two-pass rule only.  AUTO (the default): the reference's rule where the data are
    // small integers, Harris on decimal data.
    {
        const bool kernels_have_it = lu_mode_ || fast_k2_available(d_, price_blocks_ + dense_blocks_);
        bool small_integers = true;
        for (size_t e = 0; e < nnz && small_integers; ++e) small_integers = value[e] == std::nearbyint(value[e]) && std::fabs(value[e]) <= 64.0;
        for (int j = 0; j < n && small_integers; ++j) small_integers = cost2[j] == std::nearbyint(cost2[j]) && std::fabs(cost2[j]) < 1048576.0;
        for (int i = 0; i < m && small_integers; ++i) small_integers = rhs[i] == std::nearbyint(rhs[i]) && std::fabs(rhs[i]) < 1048576.0;
        if (opt_.ratio_rule == RELP_RATIO_TEXTBOOK && !kernels_have_it)
            throw std::invalid_argument("RELP_RATIO_TEXTBOOK: the reference's ratio test is implemented up to 8192 rows (the multi-workgroup ratio test has the two-pass rule only)");
        ratio_textbook_ = opt_.ratio_rule == RELP_RATIO_TEXTBOOK || (opt_.ratio_rule == RELP_RATIO_AUTO && small_integers && kernels_have_it);
    }
    // small LPs: ratio test and inverse update in one launch (pivot_fused_kernel; RELP_NO_FUSED=1 keeps the three-kernel pivot)
    fused_ = !lu_mode_ && !bounded_ && !eta_mode_ && n_dense == 0 && ftran_slices_ == 0 && !d_.track_touched && d_.ell_w == ELL_W &&
             fused_pivot_available(d_, price_blocks_) && opt_.pivot_kernels != 1;
    if (fused_) {
        for (int k = 0; k < 2; ++k) {
            d_.state[k].ctl = dmalloc<Ctl>(1);
            d_.state[k].xB = dmalloc<double>(m);
            d_.state[k].basis = dmalloc<int>(m);
        }
        ensure_polish_buffers();  // the second buffer of the out-of-place update
    }
    d_.scratch = dmalloc<double>((size_t)std::max(m, n) * 3 + 16);  // fine-grained ops carve m ints + 2 m doubles out of it
    d_.ctl = dmalloc<Ctl>(1);
    d_.dbg = dmalloc<unsigned long long>(64);
    RELP_HIP(hipMemsetAsync(d_.dbg, 0, 64 * sizeof(unsigned long long), stream_));

    upload_vec(d_.col_start, col_start, stream_);
    upload_vec(d_.row_index, row_index, stream_);
    upload_vec(d_.value, value, stream_);
    upload_vec(d_.row_start, row_start, stream_);
    upload_vec(d_.col_index, col_index, stream_);
    upload_vec(d_.row_value, row_value, stream_);
    upload_vec(d_.cost1, cost1, stream_);
    upload_vec(d_.cost2, cost2, stream_);
    upload_vec(d_.rhs, rhs, stream_);
    upload_vec(d_.rhs0, rhs, stream_);
    if (d_.ell_w == 2 && !d_.cost8) {  // columns with values: the packed records of price_kernel<.., 2>
        d_.prw = dmalloc<double>((size_t)4 * m);
        RELP_HIP(hipMemsetAsync(d_.prw, 0, (size_t)4 * m * sizeof(double), stream_));
    } else if (d_.ell_w == 2) {  // generated columns: -pi from its own vector, rho_p's non-zero rows as bits (bytes beyond LDS)
        d_.price_unit_pairs = sw(RELP_SW_PRICE_UNIT_PAIRS);
        d_.rho_words = ((m + 127) / 128) * 4;
        if ((size_t)d_.rho_words * 4 > 64 * 1024 || sw(RELP_SW_NO_RHO_BITS) || d_.price_unit_pairs) d_.rho_words = 0;
        if (d_.rho_words) {
            d_.rho_bits = dmalloc<unsigned>((size_t)2 * d_.rho_words);
            RELP_HIP(hipMemsetAsync(d_.rho_bits, 0, (size_t)2 * d_.rho_words * sizeof(unsigned), stream_));
        } else {
            d_.rho_nz = dmalloc<unsigned char>(m);
            RELP_HIP(hipMemsetAsync(d_.rho_nz, 0, m, stream_));
        }
    }
    RELP_HIP(hipMemsetAsync(d_.rho, 0, m * sizeof(double), stream_));
    RELP_HIP(hipMemsetAsync(d_.w, 0, m * sizeof(double), stream_));
    RELP_HIP(hipMemsetAsync(d_.alpha, 0, m * sizeof(double), stream_));
    RELP_HIP(hipMemsetAsync(d_.gamma, 0, n * sizeof(double), stream_));
    RELP_HIP(hipStreamSynchronize(stream_));
    configure_lds(std::min<size_t>(price_lds_, 160 * 1024 - 1024));
    tick("uploads");

    stats_.price_bytes = (long long)(col_start[n] - col_start[sparse_first_]) * 12 + (long long)(n - n_art) * 24 +
                         (long long)n_dense * m * dense_entry_bytes_;  // upper bound: every dense column non-basic
    if (d_.cost8) stats_.price_bytes = (long long)(n - n_art) * (8 + 1 + 4);  // endpoints, cost byte, pos (+ the weight of the few columns that need it)
    stats_.update_bytes = (long long)2 * m * m * 8;
    h_basis_.assign(m, -1);
    h_solution_.assign(md.nr_columns(), 0.0);
    if (lu_mode_ || opt_.crash) {
        h_col_start_ = col_start;
        h_row_index_ = row_index;
        h_value_ = value;
        h_rhs_ = rhs;
        if (opt_.crash) {  // the rows by columns too (the crash walks them)
            h_row_start_ = row_start;
            h_col_index_ = col_index;
        }
    }
}

Ctl Solver::read_ctl() {
    // Every batch of launches ends in a read of the control block (never inside a stream capture): the place where a
    // rejected launch -- a grid or an LDS request the device refuses -- surfaces instead of leaving stale device state behind.
    RELP_HIP(hipGetLastError());
    Ctl c;
    RELP_HIP(hipMemcpyAsync(&c, d_.ctl, sizeof(Ctl), hipMemcpyDeviceToHost, stream_));
    RELP_HIP(hipStreamSynchronize(stream_));
    if (c.status == ST_REFACTOR_FAILED) {
        // the refactorisation kernels gave up (a capacity, a row too long for the eliminating wave): nothing has pivoted since, the
        // basis on the device is the one they were given -- factorise it on the host (which also grows what was too small)
        ++device_refactor_failures_;
        if (opt_.verbose > 0) {
            int info[LUF_INFO_WORDS];
            RELP_HIP(hipMemcpy(info, lu().device_info(), sizeof(info), hipMemcpyDeviceToHost));
            fprintf(stderr, "[lu] the device refactorisation gave up with status %d (m %d): host fallback\n", info[LUF_STATUS], d_.m);
        }
        c.status = ST_REFACTOR;
        RELP_HIP(hipMemcpyAsync(d_.ctl, &c, sizeof(Ctl), hipMemcpyHostToDevice, stream_));
        RELP_HIP(hipStreamSynchronize(stream_));
        refactor_lu_host(true);
        refactors_--;  // (counted once, by the attempt on the device)
        RELP_HIP(hipMemcpyAsync(&c, d_.ctl, sizeof(Ctl), hipMemcpyDeviceToHost, stream_));
        RELP_HIP(hipStreamSynchronize(stream_));
    }
    return c;
}
void Solver::write_ctl(const Ctl& c) {
    RELP_HIP(hipMemcpyAsync(d_.ctl, &c, sizeof(Ctl), hipMemcpyHostToDevice, stream_));
    RELP_HIP(hipStreamSynchronize(stream_));
}

// `Tableau::<_, Partially<_>>::new` + `Carry::create_for_partially_artificial` (partially.rs:125-205, carry/mod.rs:397-442):
// B = I, basis = artificial k on its row / free slack pivot on the others, b = rhs.
void Solver::begin_phase_one() {
    if (!loaded_) throw std::runtime_error("no LP loaded");
    RELP_HIP(hipSetDevice(opt_.device));
    const MatrixData& md = form_.data;
    const int m = d_.m, n = d_.n, n_art = d_.n_art;
    std::vector<int> basis(m), pos(n, -1);
    if (bounded_)  // a variable whose two bounds coincide can never move: it is not priced (pos -3; see DeviceLP::pos)
        for (int j = n_art; j < n; ++j)
            if (zero_width_[j]) pos[j] = -3;
    auto pivots = md.pivot_element_indices();
    std::vector<int> real_column_of_row(m, -1);
    for (auto& [row, column] : pivots)
        if (row < m && column < n - n_art) real_column_of_row[row] = column;
    int k = 0;
    for (int i = 0; i < m; ++i) {
        basis[i] = real_column_of_row[i] < 0 ? k++ : n_art + real_column_of_row[i];
        pos[basis[i]] = i;
    }
    upload_vec(d_.basis, basis, stream_);
    upload_vec(d_.pos, pos, stream_);
    RELP_HIP(hipMemcpyAsync(d_.rhs, d_.rhs0, m * sizeof(double), hipMemcpyDeviceToDevice, stream_));
    if (bounded_) {  // nothing is complemented; the initial basic variables (artificials, <= slacks) have no upper bound
        std::vector<double> xub(m, std::numeric_limits<double>::infinity());
        upload_vec(d_.xub, xub, stream_);
        RELP_HIP(hipMemsetAsync(d_.flipped, 0, n * sizeof(int), stream_));
        RELP_HIP(hipStreamSynchronize(stream_));
    }
    RELP_HIP(hipMemcpyAsync(d_.xB, d_.rhs, m * sizeof(double), hipMemcpyDeviceToDevice, stream_));
    if (lu_mode_) lu_identity();
    else launch_identity(d_.Binv, m, d_.ld, stream_);
    RELP_HIP(hipMemsetAsync(d_.touched, 0, m * sizeof(int), stream_));  // every column is a unit vector
    binv_identity_ = true;
    refactors_ = 0;
    device_refactor_failures_ = 0;
    refactor_seconds_ = 0.0;
    Ctl c{};
    c.forced_q = c.forced_p = -1;
    c.last_selected = -1;
    c.scan_column = std::numeric_limits<int>::max();
    write_ctl(c);
    pivots_[0] = pivots_[1] = 0;
    polishes_ = 0;
    max_residual_ = 0.0;
    since_polish_ = 0;
    polish_scale_ = 1;
    redundant_rows_.clear();
    gamma_ready_ = false;
    if (opt_.crash && n_art > 0) crash_basis();
    set_phase(n_art > 0 ? 1 : 2);
}


// Triangular crash basis (relp_options.crash; an EXTENSION: the reference starts every row without a slack pivot on an
// artificial, partially.rs:125-205, and on the max-flow LP of examples/max_flow.rs pivots all V - 2 of them out one
// degenerate pivot at a time).  Columns that have exactly ONE entry in the rows still covered by an artificial are assigned to
// that row, breadth first -- on an incidence matrix that is a spanning forest grown from s and t.  The basis is triangular
// by construction, so its inverse comes from sparse back-substitution on the host (no factorisation) and is scattered into
// the identity that `begin_phase_one` has just written; x_B, the touched-column list and the steepest-edge weights
// gamma_j = 1 + |B^-1 a_j|^2 (pivot_rule.rs:202-219) are computed from the same sparse columns.  The crash is only kept when
// it is primal feasible; phase one then starts from it (with zero artificials left it ends without a pivot).
bool Solver::crash_basis() {
    if (lu_mode_ || eta_mode_ || d_.n_dense > 0 || h_col_start_.empty() || h_row_start_.empty()) return false;
    const int m = d_.m, n = d_.n, n_art = d_.n_art;
    const bool timing = diagnostic("RELP_TIME_SOLVE");
    double t_last = now_seconds();
    auto tick = [&](const char* what) {
        if (!timing) return;
        const double t = now_seconds();
        fprintf(stderr, "[crash] %-28s %.3f ms\n", what, (t - t_last) * 1e3);
        t_last = t;
    };
    std::vector<int> basis(m);
    RELP_HIP(hipMemcpyAsync(basis.data(), d_.basis, m * sizeof(int), hipMemcpyDeviceToHost, stream_));
    RELP_HIP(hipStreamSynchronize(stream_));
    const std::vector<int>& cs = h_col_start_;
    const std::vector<int>& ri = h_row_index_;
    const std::vector<double>& va = h_value_;
    std::vector<char> uncovered(m, 0);
    for (int i = 0; i < m; ++i) uncovered[i] = basis[i] < n_art ? 1 : 0;
    // rows -> columns: the CSR of the device LP kept by upload() (artificial columns included: skipped below)
    const std::vector<int>& row_start = h_row_start_;
    const std::vector<int>& row_cols = h_col_index_;
    std::vector<int> count(n, 0);
    for (int j = n_art; j < n; ++j)
        for (int e = cs[j]; e < cs[j + 1]; ++e) count[j] += uncovered[ri[e]];
    std::vector<int> queue;
    queue.reserve(n - n_art);
    std::vector<char> is_basic(n, 0);
    for (int i = 0; i < m; ++i) is_basic[basis[i]] = 1;
    for (int j = n_art; j < n; ++j)
        if (count[j] == 1 && !is_basic[j] && !(bounded_ && zero_width_[j])) queue.push_back(j);
    std::vector<int> order_of_row(m, -1), crash_rows;  // order in which the rows were covered
    std::vector<double> diagonal(m, 1.0);
    for (size_t head = 0; head < queue.size(); ++head) {
        const int j = queue[head];
        if (count[j] != 1 || is_basic[j]) continue;
        int r = -1;
        double pivot = 0.0, largest = 0.0;
        for (int e = cs[j]; e < cs[j + 1]; ++e) {
            largest = std::max(largest, std::fabs(va[e]));
            if (uncovered[ri[e]]) { r = ri[e]; pivot = va[e]; }
        }
        if (r < 0 || std::fabs(pivot) < 0.1 * largest) continue;  // (a small diagonal would make the triangular basis ill conditioned)
        is_basic[basis[r]] = 0;
        basis[r] = j;
        is_basic[j] = 1;
        uncovered[r] = 0;
        order_of_row[r] = (int)crash_rows.size();
        crash_rows.push_back(r);
        diagonal[r] = pivot;
        for (int e = row_start[r]; e < row_start[r + 1]; ++e) {
            const int j2 = row_cols[e];
            if (j2 < n_art) continue;
            if (--count[j2] == 1 && !is_basic[j2] && !(bounded_ && zero_width_[j2])) queue.push_back(j2);
        }
    }
    const int covered = (int)crash_rows.size();
    tick("rows by columns, BFS");
    if (covered == 0) return false;
    // Inverse by back-substitution: B (rows x positions, position of a crash column = its row) is upper triangular in the
    // order [rows that kept a unit column | crash rows in covering order].  Column r of B^-1 (r a crash row) solves B v = e_r.
    const size_t entry_cap = (size_t)64 * m + (1u << 22);
    std::vector<size_t> inv_start(covered + 1, 0);
    std::vector<int> inv_pos;
    std::vector<double> inv_val;
    std::vector<double> work(m, 0.0);
    std::vector<char> in_heap(m, 0);
    std::vector<std::pair<int, int>> heap;  // (covering order, row): max-heap
    for (int k = 0; k < covered; ++k) {
        const int r = crash_rows[k];
        work[r] = 1.0;
        heap.clear();
        heap.push_back({k, r});
        in_heap[r] = 1;
        while (!heap.empty()) {
            std::pop_heap(heap.begin(), heap.end());
            const int row = heap.back().second;
            heap.pop_back();
            in_heap[row] = 0;
            const double residual = work[row];
            work[row] = 0.0;
            if (residual == 0.0) continue;
            const double v = residual / diagonal[row];
            inv_pos.push_back(row);
            inv_val.push_back(v);
            if (order_of_row[row] < 0) continue;  // a unit column: nothing to propagate
            const int j = basis[row];
            for (int e = cs[j]; e < cs[j + 1]; ++e) {
                const int r2 = ri[e];
                if (r2 == row) continue;
                work[r2] -= va[e] * v;
                if (!in_heap[r2]) {
                    in_heap[r2] = 1;
                    heap.push_back({order_of_row[r2], r2});
                    std::push_heap(heap.begin(), heap.end());
                }
            }
        }
        inv_start[k + 1] = inv_pos.size();
        if (inv_pos.size() > entry_cap) return false;  // not a sparse inverse: leave the start to phase one
    }
    tick("sparse inverse");
    // x_B = B^-1 b, must be a basic feasible solution of the phase-one problem
    std::vector<double> xb(m, 0.0);
    for (int i = 0; i < m; ++i)
        if (order_of_row[i] < 0) xb[i] = h_rhs_[i];
    for (int k = 0; k < covered; ++k) {
        const double b = h_rhs_[crash_rows[k]];
        if (b == 0.0) continue;
        for (size_t e = inv_start[k]; e < inv_start[k + 1]; ++e) xb[inv_pos[e]] += inv_val[e] * b;
    }
    std::vector<double> ub;
    if (bounded_) {
        ub.resize(n);
        RELP_HIP(hipMemcpyAsync(ub.data(), d_.ub, n * sizeof(double), hipMemcpyDeviceToHost, stream_));
        RELP_HIP(hipStreamSynchronize(stream_));
    }
    double scale = 1.0;
    for (int i = 0; i < m; ++i) scale = std::max(scale, std::fabs(h_rhs_[i]));
    for (int i = 0; i < m; ++i) {
        if (xb[i] < -1e-9 * scale) return false;
        if (bounded_ && xb[i] > ub[basis[i]] + 1e-9 * scale) return false;
        if (xb[i] < 0.0) xb[i] = 0.0;
    }
    // steepest-edge weights of the non-basic columns from the sparse inverse columns
    std::vector<double> gamma(n, 1.0);
    if (opt_.pivot_rule == RELP_PIVOT_STEEPEST_EDGE) {
        // (independent columns: host threads, each with its own accumulator -- 41 ms on one core for the 1 M arcs of config 5)
        const int n_threads = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
        auto range = [&](int first, int last) {
            std::vector<int> touched_list, mark(m, -1);
            std::vector<double> acc(m, 0.0);
            auto add = [&](int p, double v, int j) {
                if (mark[p] != j) {
                    mark[p] = j;
                    acc[p] = 0.0;
                    touched_list.push_back(p);
                }
                acc[p] += v;
            };
            for (int j = first; j < last; ++j) {
                if (is_basic[j]) continue;
                touched_list.clear();
                for (int e = cs[j]; e < cs[j + 1]; ++e) {
                    const int r = ri[e], k = order_of_row[r];
                    if (k < 0) add(r, va[e], j);
                    else
                        for (size_t t = inv_start[k]; t < inv_start[k + 1]; ++t) add(inv_pos[t], va[e] * inv_val[t], j);
                }
                double sum = 1.0;
                for (int p : touched_list) sum += acc[p] * acc[p];
                gamma[j] = sum;
            }
        };
        const int columns = n - n_art;
        if (n_threads <= 1 || columns < 65536) {
            range(n_art, n);
        } else {
            std::vector<std::thread> pool;
            const int chunk = (columns + n_threads - 1) / n_threads;
            for (int t = 0; t < n_threads; ++t) {
                const int first = n_art + t * chunk, last = std::min(n, first + chunk);
                if (first < last) pool.emplace_back(range, first, last);
            }
            for (auto& th : pool) th.join();
        }
        std::fill(work.begin(), work.end(), 0.0);
    }
    tick("x_B, weights");
    // ---- device state -------------------------------------------------------------------------------------------
    std::vector<long long> scatter_index(inv_pos.size());
    for (int k = 0; k < covered; ++k)
        for (size_t e = inv_start[k]; e < inv_start[k + 1]; ++e) scatter_index[e] = (long long)crash_rows[k] * d_.ld + inv_pos[e];
    long long* d_index = nullptr;
    double* d_value = nullptr;
    RELP_HIP(hipMalloc(reinterpret_cast<void**>(&d_index), scatter_index.size() * sizeof(long long)));
    RELP_HIP(hipMalloc(reinterpret_cast<void**>(&d_value), inv_val.size() * sizeof(double)));
    RELP_HIP(hipMemcpyAsync(d_index, scatter_index.data(), scatter_index.size() * sizeof(long long), hipMemcpyHostToDevice, stream_));
    RELP_HIP(hipMemcpyAsync(d_value, inv_val.data(), inv_val.size() * sizeof(double), hipMemcpyHostToDevice, stream_));
    launch_scatter(d_.Binv, d_index, d_value, (long long)inv_val.size(), stream_);
    std::vector<int> pos(n, -1), touched(m, 0);
    if (bounded_)
        for (int j = n_art; j < n; ++j)
            if (zero_width_[j]) pos[j] = -3;
    for (int i = 0; i < m; ++i) pos[basis[i]] = i;
    for (int r : crash_rows) touched[r] = 1;
    upload_vec(d_.basis, basis, stream_);
    upload_vec(d_.pos, pos, stream_);
    upload_vec(d_.xB, xb, stream_);
    upload_vec(d_.gamma, gamma, stream_);
    upload_vec(d_.touched, touched, stream_);
    upload_vec(d_.tlist, crash_rows, stream_);
    if (bounded_) {
        std::vector<double> xub(m);
        for (int i = 0; i < m; ++i) xub[i] = ub[basis[i]];
        upload_vec(d_.xub, xub, stream_);
    }
    Ctl c = read_ctl();  // (also waits for the uploads)
    c.touched_count = covered;
    write_ctl(c);
    (void)hipFree(d_index);
    (void)hipFree(d_value);
    binv_identity_ = false;
    gamma_ready_ = opt_.pivot_rule == RELP_PIVOT_STEEPEST_EDGE;
    crash_rows_covered_ = covered;
    tick("uploads");
    return true;
}

// `Tableau::from_artificial` (non_artificial.rs:99-120): same basis, provider costs; -pi, -obj and the weights are
// recomputed from the resident inverse (carry/mod.rs:499-525; pivot_rule.rs:202-219).
void Solver::begin_phase_two() { set_phase(2); }

void Solver::set_phase(int phase) {
    const int phase_before = phase_;
    phase_ = phase;
    RELP_HIP(hipMemcpyAsync(d_.cost, phase == 1 ? d_.cost1 : d_.cost2, d_.n * sizeof(double), hipMemcpyDeviceToDevice, stream_));
    if (d_.cost8) {  // generated columns: the priced columns (never the artificials) cost 0 in phase one
        if (phase == 1) RELP_HIP(hipMemsetAsync(d_.cost8, 0, d_.n, stream_));
        else RELP_HIP(hipMemcpyAsync(d_.cost8, d_.cost8_2, d_.n, hipMemcpyDeviceToDevice, stream_));
    }
    if (lu_mode_) launch_lu_pi(d_, lu().device(), stream_);
    else launch_pi(d_, stream_);
    // Steepest-edge weights gamma_j = 1 + |B^-1 a_j|^2 do not depend on the costs, and the recurrences that maintain them
    // are exact: what phase one leaves is what `SteepestDescentAlongObjective::new` (pivot_rule.rs:202-219) would recompute
    // for phase two.  Recomputing costs one pass over the inverse per column pair -- fine for Netlib, 1 TB for the 1 M-arc
    // max-flow LP -- so large LPs keep the weights (after flushing the update of the last zero-level pivot).
    const double carry_threshold = opt_.carry_weights_min > 0.0 ? opt_.carry_weights_min : 4e9;  // (test hook)
    // (the LU carry always keeps them: recomputing is one FTRAN per non-basic column)
    const bool carry_weights = phase == 2 && phase_before == 1 && !binv_identity_ &&
                               (lu_mode_ || (double)(d_.n - d_.n_art) * (double)d_.m > carry_threshold);
    if (opt_.pivot_rule == RELP_PIVOT_STEEPEST_EDGE) {
        if (carry_weights) {
            Ctl pending = read_ctl();
            if (pending.pending) {
                pending.status = ST_RUNNING;
                write_ctl(pending);
                RELP_HIP(hipMemcpyAsync(d_.cost, d_.cost1, d_.n * sizeof(double), hipMemcpyDeviceToDevice, stream_));
                enqueue_price(0);  // applies the pending Goldfarb-Reid update; its candidates are discarded
                RELP_HIP(hipMemcpyAsync(d_.cost, d_.cost2, d_.n * sizeof(double), hipMemcpyDeviceToDevice, stream_));
            }
        } else if (gamma_ready_) {
            gamma_ready_ = false;  // the crash computed them on the host from its sparse inverse
        } else if (lu_mode_ && !binv_identity_) {
            launch_lu_gamma(d_, lu().device(), stream_);
        } else {
            launch_gamma_init(d_, binv_identity_ ? 1 : 0, stream_);
        }
    }
    Ctl c = read_ctl();
    const double minus_obj = c.minus_obj;
    const int touched_count = c.touched_count;
    const long long bound_flips = c.bound_flips;
    c = Ctl{};
    c.minus_obj = minus_obj;
    c.touched_count = touched_count;
    c.bound_flips = bound_flips;
    c.forced_q = c.forced_p = -1;
    c.last_selected = -1;
    c.scan_column = std::numeric_limits<int>::max();
    write_ctl(c);
}

// One batch of `count` iterations of the loop of phase_one.rs:134-178 / phase_two.rs:36-58.
void Solver::launch_pivots(int count, bool forced) {
    if (fused_ && !forced) {  // two kernels per pivot; x_B, basis and control block alternate between their two copies (kernels.hip, K23)
        launch_begin_batch(d_, count, stream_);
        for (int it = 0; it < count; ++it) {
            enqueue_price_fused(it & 1);
            enqueue_pivot_fused(it & 1);
        }
        launch_commit(d_, count & 1, stream_);
        stats_.launches += 2 + 2LL * count;
        stats_.price_launches += count;
        return;
    }
    launch_budget(d_, count, stream_);
    if (lu_mode_) {  // two kernels per pivot: the pricing pass and the single-workgroup LU kernel
        for (int it = 0; it < count; ++it) {
            enqueue_price(0);
            enqueue_ftran_ratio(0);
        }
        stats_.launches += 1 + 2LL * count;
        stats_.price_launches += count;
        return;
    }
    for (int it = 0; it < count; ++it) {
        enqueue_price(0, it == 0);
        enqueue_ftran_ratio(0);
        enqueue_update();
        if (eta_mode_ && ((it + 1) % d_.eta_cap == 0 || it + 1 == count)) enqueue_consolidate();
    }
    stats_.launches += 1 + (3LL + (dense_blocks_ > 0) + 2 * (ftran_slices_ > 0)) * count;
    stats_.price_launches += count;
}

// Pricing pass: the dense block (if any) streams through price_dense_kernel, every other column through the CSC kernel.
void Solver::enqueue_price(int skip_weights, bool first_of_batch) {
    // beside a dense block the CSC kernel only sees the short slack columns: staging -pi, rho_p, w in LDS (3 m doubles per
    // workgroup) would cost more than the gathers it saves.  (Running it on a second stream beside the dense pass was
    // measured too: the fork/join edges of the captured graph cost 15 us per pivot against the 8 us they hide.)
    // ... and beyond 4096 rows as well: a workgroup prices 32 columns and would stage 3 m doubles for them, one workgroup per CU
    // (80BAU3B, m = 5746: 486 workgroups x 138 KB = 67 MB of staging against 1.5 MB of gathers; 53.8 -> 45.7 us per pivot without).
    // Between 2000 and 2800 rows the two forms are within the run-to-run noise (BNL2, CYCLE, GREENBEA).  RELP_PRICE_LDS_MAX: A/B hook.
    const size_t lds_max = opt_.price_lds_max > 0 ? (size_t)opt_.price_lds_max : (size_t)96 * 1024;
    const bool use_lds = price_lds_ <= lds_max && dense_blocks_ == 0;
    // slack_in_btran_: the BTRAN pass of the previous pivot has priced the slack columns (weights included); only the first
    // pivot of a batch has no predecessor in the batch, and its pass must not apply the weight update a second time
    if (price_blocks_ > 0 && (!slack_in_btran_ || first_of_batch))
        launch_price(d_, opt_.pivot_rule, price_blocks_, use_lds ? price_lds_ : 0, use_lds, slack_in_btran_ ? 1 : skip_weights, opt_.tol_dual,
                     sparse_first_, d_.n, 0, stream_);
    if (dense_blocks_ > 0) launch_price_dense(d_, dense_blocks_, skip_weights, opt_.tol_dual, price_blocks_, stream_);
}

// Fused mode: the pricing pass before pivot k reads the control block of copy k & 1 (it writes nothing of the twin state).
void Solver::enqueue_price_fused(int parity) {
    DeviceLP d = d_;
    d.ctl = d_.state[parity].ctl;
    const bool use_lds = price_lds_ <= 160 * 1024 - 1024;
    launch_price(d, opt_.pivot_rule, price_blocks_, use_lds ? price_lds_ : 0, use_lds, 0, opt_.tol_dual, sparse_first_, d_.n, 0, stream_);
}
void Solver::enqueue_pivot_fused(int parity) {
    launch_pivot_fused(d_, opt_.pivot_rule, parity, price_blocks_, opt_.tol_pivot, ratio_delta(), phase_ == 2 ? 1 : 0, stream_);
}

// The basis update: rank-one update of the explicit inverse (K3), or -- deferred product form -- the eta bookkeeping plus
// one read-only pass for rho_p, w and -pi.
void Solver::enqueue_update() {
    if (eta_mode_) launch_eta_update(d_, opt_.tol_dual, stream_);
    else launch_update(d_, stream_);
}
// Fold the pending etas into the stored inverse (no-op kernels when there are none; runs whatever the status is, so that
// everything outside the pivot loop sees the plain explicit inverse).
void Solver::enqueue_consolidate() {
    if (eta_mode_) launch_eta_consolidate(d_, stream_);
}

// Entering column + FTRAN + ratio test (+ updates in mode 0).  Long (dense) columns take the multi-block FTRAN.
void Solver::enqueue_ftran_ratio(int mode) {
    const int skip_art = phase_ == 2 ? 1 : 0;
    const int slots = price_blocks_ + dense_blocks_;
    if (lu_mode_) {
        hipEvent_t start = nullptr, stop = nullptr;
        take_launch_timer(1, &start, &stop);
        launch_lu_pivot(d_, lu().device(), opt_.pivot_rule, slots, opt_.tol_pivot, ratio_delta(), skip_art, mode, refactor_period_, stream_, start, stop);
        return;
    }
    if (ftran_slices_ > 0) {
        launch_ftran_partial(d_, ftran_slices_, slots, opt_.pivot_rule, stream_);
        launch_alpha_reduce(d_, ftran_slices_, stream_);
    }
    launch_ftran_ratio(d_, opt_.pivot_rule, slots, opt_.tol_pivot, ratio_delta(), skip_art, mode, ftran_slices_ > 0 ? 1 : 0, stream_);
}

void Solver::destroy_graphs() {
    for (int k = 0; k < 4; ++k) {
        if (graph_exec_[k]) { (void)hipGraphExecDestroy(graph_exec_[k]); graph_exec_[k] = nullptr; }
        if (graph_[k]) { (void)hipGraphDestroy(graph_[k]); graph_[k] = nullptr; }
        graph_count_[k] = 0;
    }
}

void Solver::build_graph(int count) {
    const int k = graph_index();
    if (graph_exec_[k] && graph_count_[k] == count) return;
    if (graph_exec_[k]) { (void)hipGraphExecDestroy(graph_exec_[k]); graph_exec_[k] = nullptr; }
    if (graph_[k]) { (void)hipGraphDestroy(graph_[k]); graph_[k] = nullptr; }
    long long launches = stats_.launches, price_launches = stats_.price_launches;
    RELP_HIP(hipStreamBeginCapture(stream_, hipStreamCaptureModeRelaxed));
    launch_pivots(count);
    RELP_HIP(hipStreamEndCapture(stream_, &graph_[k]));
    RELP_HIP(hipGraphInstantiate(&graph_exec_[k], graph_[k], nullptr, nullptr, 0));
    stats_.launches = launches;
    stats_.price_launches = price_launches;
    graph_count_[k] = count;
}

// Newton-Schulz polish (see kernels.hip).  Two iterations at most; the residual before the polish is recorded.
void Solver::polish(bool refresh_vectors, bool force) {
    if (lu_mode_) {  // the LU carry's refresh is a refactorisation
        refactor_lu(refresh_vectors);
        return;
    }
    // nothing has changed since the inverse was last made exact (identity, crash basis, set_basis, the previous polish): at
    // m = 65 534 the residual pass and the two refresh passes are 34 GB each
    if (since_polish_ == 0 && !force && !(opt_.switches & RELP_SW_POLISH_ALWAYS)) return;
    const int m = d_.m;
    // dense pipeline: only the columns of the stored inverse that are not unit vectors take part (the corresponding
    // rows of S are zero and those columns of the polished inverse do not change): both GEMMs shrink by m / touched
    const bool by_rows = d_.track_touched && gemm_row_lists_supported();
    const int* rows = by_rows ? d_.tlist : nullptr;
    int n_rows = m;
    if (by_rows) n_rows = read_ctl().touched_count;
    if (d_.n_dense > 0) ensure_polish_buffers();  // the dense residual is a GEMM into R through Binv2
    for (int it = 0; it < 2; ++it) {
        RELP_HIP(hipMemsetAsync(&d_.ctl->residual, 0, sizeof(double), stream_));
        if (by_rows && d_.R) RELP_HIP(hipMemsetAsync(d_.R, 0, (size_t)m * d_.ld * sizeof(double), stream_));
        if (d_.n_dense > 0) launch_residual_dense(d_, d_.Binv2, d_.Binv, d_.R, rows, n_rows, stream_);
        else launch_residual(d_, d_.Binv, d_.R, stream_);  // R == nullptr: norm only
        Ctl c = read_ctl();  // max |I - B' T|
        if (d_.R == nullptr && c.residual >= 1e-12 && c.residual < 0.5) {  // something to correct: now S itself is needed
            ensure_polish_buffers();
            if (by_rows) RELP_HIP(hipMemsetAsync(d_.R, 0, (size_t)m * d_.ld * sizeof(double), stream_));
            RELP_HIP(hipMemsetAsync(&d_.ctl->residual, 0, sizeof(double), stream_));
            launch_residual(d_, d_.Binv, d_.R, stream_);
            c = read_ctl();
        }
        if (!(c.residual == c.residual)) throw std::runtime_error("NaN in basis inverse");
        if (it == 0) {
            max_residual_ = std::max(max_residual_, c.residual);
            // the drift seen after this many pivots schedules the next polish: far below the working accuracy -> wait
            // twice as long (up to 16 periods), close to it -> back to the configured period
            // (an inverse that is still EXACT, as on totally unimodular bases, has not drifted at all: four times as long, up to 256)
            if (c.residual == 0.0) polish_scale_ = std::min(256, polish_scale_ * 4);
            else if (c.residual < 1e-8) polish_scale_ = std::min(std::max(16, polish_scale_), polish_scale_ * 2);
            else if (c.residual > 1e-6) polish_scale_ = 1;
        }
        if (c.residual < 1e-12) break;  // nothing to correct: skip the second GEMM
        if (c.residual >= 0.5) {  // drifted too far for the quadratic iteration: rebuild from the basis columns
            invert_from_scratch();
            break;
        }
        launch_gemm_polish(d_.Binv, d_.R, d_.Binv2, m, d_.ld, rows, n_rows, stream_);
        if (by_rows) launch_copy_rows(d_.Binv2, d_.Binv, m, d_.ld, rows, n_rows, stream_);
        else RELP_HIP(hipMemcpyAsync(d_.Binv, d_.Binv2, (size_t)m * d_.ld * sizeof(double), hipMemcpyDeviceToDevice, stream_));
        if (c.residual < 1e-8) break;  // one quadratic step takes it to ~residual^2
    }
    binv_identity_ = false;
    polishes_++;
    since_polish_ = 0;
    if (refresh_vectors) {  // pi_kernel also rewrites minus_obj from the refreshed xB; everything stays on the stream
        launch_xb(d_, stream_);
        launch_pi(d_, stream_);
    }
}

// From-scratch inverse by Newton-Schulz from X0 = B' / (|B|_1 |B|_inf) (converges for every nonsingular B).
// Plays the role of `BasisInverse::invert` (lower_upper/mod.rs:78-92) for `from_basis` / warm starts.
void Solver::ensure_polish_buffers() {
    if (d_.Binv2 == nullptr) d_.Binv2 = dmalloc<double>((size_t)d_.m * d_.ld);
    if (d_.R == nullptr) d_.R = dmalloc<double>((size_t)d_.m * d_.ld);
}

void Solver::invert_from_scratch() {
    if (lu_mode_) {
        refactor_lu(false);
        return;
    }
    const int m = d_.m;
    ensure_polish_buffers();
    std::vector<int> basis(m);
    RELP_HIP(hipMemcpyAsync(basis.data(), d_.basis, m * sizeof(int), hipMemcpyDeviceToHost, stream_));
    RELP_HIP(hipStreamSynchronize(stream_));
    const MatrixData& md = form_.data;
    std::vector<double> row_sum(m, 0.0);
    double norm1 = 0.0;
    for (int k = 0; k < m; ++k) {
        double col_sum = 0.0;
        if (basis[k] < d_.n_art) {
            col_sum = 1.0;  // identity column; its row is found below through pos/CSR, the bound is enough here
        } else {
            SparseColumn c = md.column(basis[k] - d_.n_art);
            for (size_t e = 0; e < c.nnz(); ++e) {
                if (c.index[e] >= m) continue;  // the bound-row entry of a bounded column (implicit bounds)
                double v = std::fabs(c.value[e].to_double());
                col_sum += v;
                row_sum[c.index[e]] += v;
            }
        }
        norm1 = std::max(norm1, col_sum);
    }
    double norm_inf = 1.0;
    for (double v : row_sum) norm_inf = std::max(norm_inf, v);
    launch_scaled_basis(d_, d_.Binv, 1.0 / (norm1 * norm_inf), stream_);
    double previous = std::numeric_limits<double>::infinity();
    for (int it = 0; it < 200; ++it) {
        RELP_HIP(hipMemsetAsync(&d_.ctl->residual, 0, sizeof(double), stream_));
        launch_residual(d_, d_.Binv, d_.R, stream_);
        launch_gemm_polish(d_.Binv, d_.R, d_.Binv2, m, d_.ld, nullptr, m, stream_);
        RELP_HIP(hipMemcpyAsync(d_.Binv, d_.Binv2, (size_t)m * d_.ld * sizeof(double), hipMemcpyDeviceToDevice, stream_));
        Ctl c = read_ctl();
        if (c.residual < 1e-11) break;
        if (it > 60 && c.residual >= previous) break;
        previous = c.residual;
    }
    launch_mark_all_touched(d_, stream_);  // no column of a fresh inverse is known to be a unit vector
    binv_identity_ = false;
}

// `InverseMaintainer::from_basis` (carry/mod.rs:444-478) + `Tableau::new_with_inverse_maintainer`: phase two from a given basis.
void Solver::set_basis(const int* basis_columns) {
    if (!loaded_) throw std::runtime_error("no LP loaded");
    RELP_HIP(hipSetDevice(opt_.device));
    const int m = d_.m, n = d_.n;
    std::vector<int> basis(m), pos(n, -1);
    if (!bounded_) {
        for (int i = 0; i < m; ++i) {
            int c = basis_columns[i];
            int dev = c >= 0 ? d_.n_art + c : (-1 - c);
            if (dev < 0 || dev >= n || pos[dev] >= 0) throw std::invalid_argument("bad basis");
            basis[i] = dev;
            pos[dev] = i;
        }
    } else {
        // Implicit bounds: `basis_columns` is a basis of the reference's formulation (one column per row of MatrixData).  A
        // bounded variable whose bound slack is NOT basic sits at its bound: non-basic and complemented on the device; with
        // the slack basic it is basic here exactly when it is basic there.
        const MatrixData& md = form_.data;
        const int rows_full = md.nr_rows(), cols_full = md.nr_columns(), n_dev = md.col_end[3];
        std::vector<int> row_of(cols_full, -1);
        std::vector<int> artificial_row_of(d_.n_art, -1);
        for (int i = 0; i < rows_full; ++i) {
            const int c = basis_columns[i];
            if (c >= 0) {
                if (c >= cols_full || row_of[c] >= 0) throw std::invalid_argument("bad basis");
                row_of[c] = i;
            } else {
                const int k = -1 - c;
                if (k >= d_.n_art || artificial_row_of[k] >= 0) throw std::invalid_argument("bad basis");
                artificial_row_of[k] = i;
            }
        }
        std::vector<int> flipped(n, 0);
        auto bound_pair = [&](int variable, int slack) {
            if (row_of[slack] < 0) {
                if (row_of[variable] < 0) throw std::invalid_argument("bad basis: neither a bounded variable nor its bound slack is basic");
                flipped[d_.n_art + variable] = 1;  // at its upper bound
                row_of[variable] = -2;             // not basic on the device
            }
        };
        for (int k2 = 0; k2 < (int)md.bound_to_variable.size(); ++k2) bound_pair(md.bound_to_variable[k2], md.col_end[3] + k2);
        for (int k2 = 0; k2 < md.nr_range; ++k2) bound_pair(md.col_end[0] + k2, md.col_end[4] + k2);
        std::vector<int> device_basic;  // device column, preferred row
        std::vector<int> wanted_row;
        for (int k = 0; k < d_.n_art; ++k)
            if (artificial_row_of[k] >= 0) { device_basic.push_back(k); wanted_row.push_back(artificial_row_of[k]); }
        for (int c = 0; c < n_dev; ++c)
            if (row_of[c] >= 0) { device_basic.push_back(d_.n_art + c); wanted_row.push_back(row_of[c]); }
        if ((int)device_basic.size() != m) throw std::invalid_argument("bad basis: it does not reduce to a basis of the constraint rows");
        std::fill(basis.begin(), basis.end(), -1);
        std::vector<int> homeless;
        for (size_t t = 0; t < device_basic.size(); ++t) {
            if (wanted_row[t] < m && basis[wanted_row[t]] < 0) basis[wanted_row[t]] = device_basic[t];
            else homeless.push_back(device_basic[t]);
        }
        size_t next = 0;
        for (int i = 0; i < m; ++i)
            if (basis[i] < 0) basis[i] = homeless[next++];
        for (int j = 0; j < n; ++j) {
            if (zero_width_[j]) flipped[j] = 0;  // both bounds are the same point
            pos[j] = zero_width_[j] ? -3 : (flipped[j] ? -2 : -1);
        }
        for (int i = 0; i < m; ++i) pos[basis[i]] = i;
        // right-hand side with the complemented columns moved over, bounds of the basic variables
        std::vector<double> ub(n), rhs(m), xub(m);
        RELP_HIP(hipMemcpyAsync(ub.data(), d_.ub, n * sizeof(double), hipMemcpyDeviceToHost, stream_));
        RELP_HIP(hipMemcpyAsync(rhs.data(), d_.rhs0, m * sizeof(double), hipMemcpyDeviceToHost, stream_));
        RELP_HIP(hipStreamSynchronize(stream_));
        for (int j = d_.n_art; j < n; ++j) {
            if (!flipped[j]) continue;
            SparseColumn column = md.column(j - d_.n_art);
            for (size_t e = 0; e < column.nnz(); ++e)
                if (column.index[e] < m) rhs[column.index[e]] -= ub[j] * column.value[e].to_double();
        }
        for (int i = 0; i < m; ++i) xub[i] = ub[basis[i]];
        upload_vec(d_.flipped, flipped, stream_);
        upload_vec(d_.rhs, rhs, stream_);
        upload_vec(d_.xub, xub, stream_);
        RELP_HIP(hipStreamSynchronize(stream_));
    }
    upload_vec(d_.basis, basis, stream_);
    upload_vec(d_.pos, pos, stream_);
    Ctl c{};
    c.forced_q = c.forced_p = -1;
    c.last_selected = -1;
    c.scan_column = std::numeric_limits<int>::max();
    write_ctl(c);
    invert_from_scratch();
    if (lu_mode_) launch_lu_xb(d_, lu().device(), stream_);
    else launch_xb(d_, stream_);
    binv_identity_ = false;
    refactors_ = 0;
    refactor_seconds_ = 0.0;
    pivots_[0] = pivots_[1] = 0;
    polishes_ = 0;
    max_residual_ = 0.0;
    since_polish_ = 0;
    polish_scale_ = 1;
    redundant_rows_.clear();
    set_phase(2);
}

long long Solver::iterate(long long count, int* stop_reason) {
    if (phase_ == 0) throw std::runtime_error("no phase started");
    RELP_HIP(hipSetDevice(opt_.device));
    long long done = 0;
    int reason = ST_BUDGET;
    long long iters_before = read_ctl().iters;  // one control-word read per batch: the next batch starts where this one ended
    // LU carry: a batch is one refactorisation cycle (period updates + the pivot that asks for the refactorisation)
    // (the refactorisation beside the pivots needs the host's attention more often than once per cycle: batches of 16 pivots)
    const int full_batch = async_refactor_ ? (opt_.pivots_per_launch > 0 && opt_.pivots_per_launch < 64 ? opt_.pivots_per_launch : 16) : lu_mode_ ? refactor_period_ + 1 : std::max(1, opt_.pivots_per_launch);
    while (done < count) {
        if (async_in_flight_ && hipEventQuery(ev_refactored_) == hipSuccess) finish_async_refactor(iters_before);
        long long room = (!lu_mode_ && opt_.polish_period > 0) ? (long long)opt_.polish_period * polish_scale_ - since_polish_ : count;
        if (room <= 0) { polish(true); continue; }  // (a polish does not touch the iteration counter)
        int batch = (int)std::min<long long>({count - done, room, (long long)full_batch});
        if (opt_.use_graph && batch == full_batch) {
            build_graph(batch);
            RELP_HIP(hipGraphLaunch(graph_exec_[graph_index()], stream_));
            stats_.launches += 1 + (lu_mode_ ? 2LL : 3LL) * batch;
            stats_.price_launches += batch;
        } else {
            launch_pivots(batch);
        }
        const long long fallbacks_before = device_refactor_failures_;
        Ctl after = read_ctl();  // (a refactorisation the kernels gave up on is redone by the host in here: nothing pivoted, go on)
        const bool fell_back = device_refactor_failures_ != fallbacks_before;
        long long made = after.iters - iters_before;
        iters_before = after.iters;
        done += made;
        since_polish_ += made;
        pivots_[phase_ - 1] += made;
        if (made > 0) binv_identity_ = false;  // (the phase hand-over must not take the weights of the identity basis)
        if (after.status == ST_NO_ENTERING || after.status == ST_UNBOUNDED) { reason = after.status; break; }
        if (after.status == ST_REFACTOR) {  // LU carry: should_refactor (or an unstable update) -- BasisInverse::invert
            // (the new factors may be on their way on the other stream: wait for them, replay, go on; else factorise now)
            if (async_in_flight_ && finish_async_refactor(after.iters)) launch_clear_refactor_status(d_, stream_);
            else refactor_lu(true, /*settle=*/false);  // (the next batch ends in read_ctl)
            continue;
        }
        // the next factors are started early enough that the pivots made meanwhile fit the log of their etas
        if (async_refactor_ && !async_in_flight_ && after.status == ST_RUNNING && lu().device().inverse_factors == 4 &&
            since_polish_ + LU_LOG_CAPACITY + full_batch > refactor_period_)
            start_async_refactor(after.iters);
        if (made == 0 && after.status == ST_RUNNING && !fell_back) break;  // defensive: nothing happened
    }
    if (stop_reason) *stop_reason = reason;
    return done;
}

// phase_one.rs:232-278.  Returns the number of redundant rows (artificials that cannot be pivoted out).
int Solver::drive_out_artificials() {
    const int m = d_.m;
    std::vector<int> basis(m);
    RELP_HIP(hipMemcpyAsync(basis.data(), d_.basis, m * sizeof(int), hipMemcpyDeviceToHost, stream_));
    RELP_HIP(hipStreamSynchronize(stream_));
    int redundant = 0;
    for (int r = 0; r < m; ++r) {
        if (basis[r] >= d_.n_art) continue;
        Ctl c = read_ctl();
        c.scan_column = std::numeric_limits<int>::max();
        write_ctl(c);
        if (lu_mode_) {  // row r of the inverse by one BTRAN, then the scan of rho_r a_j over the non-basic columns
            double* rowvec = d_.scratch + (d_.m + 1) / 2 + 1 + d_.m;
            int* d_slot = reinterpret_cast<int*>(d_.scratch);
            double* d_one = d_.scratch + (d_.m + 1) / 2 + 1;
            const double one = 1.0;
            RELP_HIP(hipMemcpyAsync(d_slot, &r, sizeof(int), hipMemcpyHostToDevice, stream_));
            RELP_HIP(hipMemcpyAsync(d_one, &one, sizeof(double), hipMemcpyHostToDevice, stream_));
            launch_lu_btran(lu().device(), d_slot, d_one, 1, rowvec, stream_);
            launch_lu_row_scan(d_, rowvec, 1e-7, stream_);
        } else {
            launch_row_scan(d_, r, 1e-7, stream_);
        }
        c = read_ctl();
        if (c.scan_column == std::numeric_limits<int>::max()) {
            redundant_rows_.push_back(r);
            ++redundant;
            continue;
        }
        c.forced_q = c.scan_column;
        c.forced_p = r;
        c.status = ST_RUNNING;
        write_ctl(c);
        // a zero-level pivot: the artificial that leaves IS zero (the phase-one objective vanished); whatever residue f64 left
        // in x_B[r] must not be divided by a small pivot element and spread over x_B
        RELP_HIP(hipMemsetAsync(d_.xB + r, 0, sizeof(double), stream_));
        launch_pivots(1, true);
        Ctl after = read_ctl();
        if (after.iters == c.iters) throw std::runtime_error("zero-level pivot failed");
        pivots_[0] += 1;
        since_polish_ += 1;
        if (after.status == ST_REFACTOR) refactor_lu(true);
    }
    return redundant;
}

void Solver::solve(relp_result* result) {
    if (!loaded_) throw std::runtime_error("no LP loaded");
    RELP_HIP(hipSetDevice(opt_.device));
    const double t0 = now_seconds();
    relp_result res{};
    exact_objective.clear();
    exact_primal.reset();
    const bool timing = diagnostic("RELP_TIME_SOLVE");  // diagnostic: host-side timeline of one solve
    double t_last = t0;
    auto tick = [&](const char* what) {
        if (!timing) return;
        RELP_HIP(hipStreamSynchronize(stream_));
        const double t = now_seconds();
        fprintf(stderr, "[solve] %-24s %.3f ms\n", what, (t - t_last) * 1e3);
        t_last = t;
    };
    begin_phase_one();
    tick("begin_phase_one");
    // 0: a safety cap far above anything a terminating solve needs (GREENBEA: 8349 pivots with m + n = 8776), so that an LP
    // that cycles in f64 comes back as RELP_RESULT_ITERATION_LIMIT instead of never
    const long long cap = opt_.max_pivots > 0 ? opt_.max_pivots : 200LL * (d_.m + d_.n) + 100000;
    int kind = RELP_RESULT_NONE;
    if (phase_ == 1) {
        int reason = 0;
        long long done = iterate(cap, &reason);
        (void)done;
        tick("phase one loop");
        if (reason == ST_UNBOUNDED) throw std::runtime_error("Artificial cost can not be unbounded.");  // phase_one.rs:151
        if (reason == ST_BUDGET) kind = RELP_RESULT_ITERATION_LIMIT;
        if (kind == RELP_RESULT_NONE) {
            polish(true);
            tick("polish");
            Ctl c = read_ctl();
            std::vector<double> xb(d_.m);
            RELP_HIP(hipMemcpyAsync(xb.data(), d_.xB, d_.m * sizeof(double), hipMemcpyDeviceToHost, stream_));
            RELP_HIP(hipStreamSynchronize(stream_));
            double scale = 1.0;
            for (double v : xb) scale += std::fabs(v);
            if (-c.minus_obj > opt_.tol_feasible * scale) {
                kind = RELP_RESULT_INFEASIBLE;  // phase_one.rs:171-173
            } else {
                tick("feasibility check");
                drive_out_artificials();
                tick("drive out artificials");
                set_phase(2);
                tick("set_phase(2)");
            }
        }
    }
    if (kind == RELP_RESULT_NONE) {
        int reason = 0;
        iterate(cap - pivots_[0], &reason);
        tick("phase two loop");
        if (reason == ST_UNBOUNDED) kind = RELP_RESULT_UNBOUNDED;  // phase_two.rs:53
        else if (reason == ST_NO_ENTERING) {
            kind = RELP_RESULT_FINITE_OPTIMUM;
            polish(true);
            tick("final polish");
        } else kind = RELP_RESULT_ITERATION_LIMIT;
    }
    // results: Carry::current_bfs (carry/mod.rs:636-645) + reconstruct_solution (matrix_data.rs:402-411)
    const int m = d_.m;
    std::vector<int> basis(m);
    std::vector<double> xb(m);
    RELP_HIP(hipMemcpyAsync(basis.data(), d_.basis, m * sizeof(int), hipMemcpyDeviceToHost, stream_));
    RELP_HIP(hipMemcpyAsync(xb.data(), d_.xB, m * sizeof(double), hipMemcpyDeviceToHost, stream_));
    Ctl c = read_ctl();
    for (int i = 0; i < m; ++i)  // never index host arrays with an unchecked device value
        if (basis[i] < 0 || basis[i] >= d_.n) throw std::runtime_error("the device returned an invalid basis (row " + std::to_string(i) + ")");
    std::fill(h_solution_.begin(), h_solution_.end(), 0.0);
    if (!bounded_) {
        for (int i = 0; i < m; ++i) {
            h_basis_[i] = basis[i] >= d_.n_art ? basis[i] - d_.n_art : -1 - basis[i];
            if (basis[i] >= d_.n_art) h_solution_[basis[i] - d_.n_art] = xb[i];
        }
    } else {
        // back to the reference's formulation: values of complemented variables are u_j - x'_j, a complemented non-basic
        // variable sits at its upper bound, and the basis of the full MatrixData has, on every bound row, the bound slack
        // (variable below its bound) or the variable itself (variable at its bound).
        const MatrixData& md = form_.data;
        std::vector<int> flipped(d_.n), pos(d_.n);
        std::vector<double> ub(d_.n);
        RELP_HIP(hipMemcpyAsync(flipped.data(), d_.flipped, d_.n * sizeof(int), hipMemcpyDeviceToHost, stream_));
        RELP_HIP(hipMemcpyAsync(pos.data(), d_.pos, d_.n * sizeof(int), hipMemcpyDeviceToHost, stream_));
        RELP_HIP(hipMemcpyAsync(ub.data(), d_.ub, d_.n * sizeof(double), hipMemcpyDeviceToHost, stream_));
        RELP_HIP(hipStreamSynchronize(stream_));
        resolve_fixed_columns(pos);
        h_basis_ = explicit_basis(basis, pos);
        for (int i = 0; i < m; ++i) {
            const int dev = basis[i];
            if (dev >= d_.n_art) h_solution_[dev - d_.n_art] = flipped[dev] ? ub[dev] - xb[i] : xb[i];
        }
        for (int j = d_.n_art; j < d_.n; ++j)
            if (pos[j] == -2) h_solution_[j - d_.n_art] = ub[j];
        const int nb = (int)md.bound_to_variable.size();
        for (int k2 = 0; k2 < nb; ++k2) {  // VariableBound rows: the bound slack of a variable below its bound
            const int j = md.bound_to_variable[k2];
            if (pos[d_.n_art + j] != -2) h_solution_[md.col_end[3] + k2] = ub[d_.n_art + j] - h_solution_[j];
        }
        for (int k2 = 0; k2 < md.nr_range; ++k2) {  // SlackBound rows (range slacks)
            const int j = md.col_end[0] + k2;
            if (pos[d_.n_art + j] != -2) h_solution_[md.col_end[4] + k2] = ub[d_.n_art + j] - h_solution_[j];
        }
    }
    res.kind = kind;
    res.pivots_phase_one = pivots_[0];
    res.pivots_phase_two = pivots_[1];
    res.polishes = polishes_;
    res.max_residual = max_residual_;
    res.refactors = refactors_;
    res.refactor_seconds = refactor_seconds_;
    res.objective = (kind == RELP_RESULT_FINITE_OPTIMUM) ? -c.minus_obj + form_.fixed_cost.to_double()
                                                         : std::numeric_limits<double>::quiet_NaN();
    tick("results");
    res.solve_seconds = now_seconds() - t0;
    // The exact certificate: optimality of the final basis; for the two other verdicts of `OptimizationResult`
    // (algorithm/mod.rs:43-47), which the reference decides exactly, a Farkas certificate / an unbounded ray.
    unbounded_column_ = kind == RELP_RESULT_UNBOUNDED ? c.q - d_.n_art : -1;
    if (opt_.certify && !bounded_ &&
        (kind == RELP_RESULT_FINITE_OPTIMUM || kind == RELP_RESULT_INFEASIBLE || kind == RELP_RESULT_UNBOUNDED)) certify(&res);
    else if (opt_.certify && kind == RELP_RESULT_FINITE_OPTIMUM) certify(&res);
    last_result = res;
    if (result) *result = res;
}

// Implicit bounds: the basis of the reference's formulation (all rows of MatrixData) that the device state stands for.  On
// every bound row the bound slack is basic when the variable is below its bound and the variable itself when it sits at it.
// A fixed variable (pos -3) sits at both of its bounds at once; which of the two the reference's formulation should see is
// decided by its reduced cost, exactly as the bound flips of a zero step used to do: negative -> "at the upper bound" (the
// variable itself basic on its bound row), else the bound slack.  Rewrites the -3 entries of `pos` to -2 / -1.
void Solver::resolve_fixed_columns(std::vector<int>& pos) {
    bool any = false;
    for (int j = d_.n_art; j < d_.n && !any; ++j) any = pos[j] == -3;
    if (!any) return;
    std::vector<double> cbar(d_.n);
    relative_costs(cbar.data());
    for (int j = d_.n_art; j < d_.n; ++j)
        if (pos[j] == -3) pos[j] = cbar[j] < 0.0 ? -2 : -1;
}

std::vector<int> Solver::explicit_basis(const std::vector<int>& basis, const std::vector<int>& pos) const {
    const MatrixData& md = form_.data;
    std::vector<int> out(md.nr_rows(), -1);
    for (int i = 0; i < d_.m; ++i) out[i] = basis[i] >= d_.n_art ? basis[i] - d_.n_art : -1 - basis[i];
    const int nb = (int)md.bound_to_variable.size();
    for (int k2 = 0; k2 < nb; ++k2) {
        const int j = md.bound_to_variable[k2];
        out[md.row_end[3] + k2] = pos[d_.n_art + j] == -2 ? j : md.col_end[3] + k2;
    }
    for (int k2 = 0; k2 < md.nr_range; ++k2) {
        const int j = md.col_end[0] + k2;
        out[md.row_end[4] + k2] = pos[d_.n_art + j] == -2 ? j : md.col_end[4] + k2;
    }
    return out;
}

void Solver::certify(relp_result* result) {
    const double t0 = now_seconds();
    bool ok = false;
    long long repairs = 0;
    std::string message;
    try {
        const int mode = result->kind == RELP_RESULT_INFEASIBLE ? 1 : result->kind == RELP_RESULT_UNBOUNDED ? 2 : 0;
        certify_basis(form_, h_basis_, opt_.device, stream_, &exact_objective, &ok, &repairs, &message, mode, unbounded_column_, &exact_primal, &certify_scratch_);
    } catch (const RatOverflow& e) {  // the f64 result stands; it is reported uncertified with the reason
        ok = false;
        message = std::string("exact certificate: ") + e.what();
    }
    result->certified = ok ? 1 : 0;
    result->exact_repair_pivots = repairs;
    result->certify_seconds = now_seconds() - t0;
    if (!ok) last_error = message;
}

// ---- LU carry ---------------------------------------------------------------------------------------
// `BasisInverse::identity` (lower_upper/mod.rs:67-76) for the start of phase one.
// A factorisation on its way on the second stream is given up (its basis is superseded): the handle's stream waits for it before the
// snapshot buffers and the other set of factor arrays are written again -- the next start_async_refactor would otherwise copy a new
// basis into d_basis_snapshot_ while the abandoned kernels still read the old one (advisor, round 5).
void Solver::abandon_async_flight() {
    if (async_in_flight_) RELP_HIP(hipStreamWaitEvent(stream_, ev_refactored_, 0));
    async_in_flight_ = false;
}
void Solver::lu_identity() {
    abandon_async_flight();
    const int m = d_.m;
    HostLU f;
    f.m = m;
    f.rowpos.resize(m);
    f.colpos.resize(m);
    for (int i = 0; i < m; ++i) f.rowpos[i] = f.colpos[i] = i;
    f.l_start.assign(m + 1, 0);
    f.u_start.assign(m + 1, 0);
    f.diag.assign(m, 1.0);
    // (device refactorisation: every array sized by bounds first, so that the layout -- and the captured graphs -- stay put)
    if (device_refactor_ && lu().prepare_device(m, refactor_period_ + 1, true, (size_t)h_col_start_.back())) destroy_graphs();
    if (lu().upload(f, refactor_period_ + 1, stream_, lu_inverse_)) destroy_graphs();  // the captured batches hold the old addresses
}
// `BasisInverse::invert(basis columns)` (lower_upper/mod.rs:78-92; called by `Carry::change_basis` when `should_refactor`,
// carry/mod.rs:584-591): Markowitz factorisation of the current basis on the host, one upload, and -- `refresh_vectors` -- x_B,
// -pi and the objective recomputed from the fresh factors (what the explicit carry's polish does too).
void Solver::refactor_lu(bool refresh_vectors, bool settle) {
    abandon_async_flight();  // (factors on their way on the other stream belong to a basis this call supersedes: never swapped in)
    if (!device_refactor_) {
        refactor_lu_host(refresh_vectors);
        return;
    }
    // Round 4: factorisation, inversion of the two triangles and the slot records are kernels on this handle's stream -- no basis
    // read-back, no upload, the host only enqueues (its time below is launch overhead).  What the kernels cannot take comes back
    // as ST_REFACTOR_FAILED at the next read of the control block (read_ctl: host fallback).
    const double t0 = now_seconds();
    LuFactorSource src;
    src.col_start = d_.col_start;
    src.row_index = d_.row_index;
    src.value = d_.value;
    src.basis = d_.basis;
    src.flipped = bounded_ ? d_.flipped : nullptr;
    const double threshold = opt_.lu_pivot_threshold > 0.0 ? opt_.lu_pivot_threshold : 0.1;
    // (dense tail: the last rows through a dense LU out of LDS.  It saves the factorisation its slowest rounds but makes the ends of both
    //  triangles dense, and the INVERTED triangles pay for that -- more entries per product and a serial chain in the inversion)
    const int dense_tail = opt_.luf_dense_tail > 0 ? opt_.luf_dense_tail : (opt_.luf_dense_tail < 0 ? 0 : 8);
    lu().refactor_device(src, threshold, 0, dense_tail, d_.ctl, ST_REFACTOR_FAILED, stream_);
    binv_identity_ = false;
    if (refresh_vectors) {
        launch_lu_xb(d_, lu().device(), stream_);
        launch_lu_pi(d_, lu().device(), stream_);
    }
    launch_clear_refactor_status(d_, stream_);
    refactors_++;
    since_polish_ = 0;
    refactor_seconds_ += now_seconds() - t0;
    // Kernels that gave up leave ST_REFACTOR_FAILED in the control block and half-written factors behind.  The pivot loop reads the block
    // after its next batch (whose kernels do nothing under that status); every other caller hands control back to code that may read x_B,
    // -pi or solve with the factors directly, so the host fallback of read_ctl runs before it does.
    if (settle) read_ctl();
}
// Round 5: `BasisInverse::invert` beside the pivots.  The basis is copied as it stands (stream-ordered behind the pivots made so
// far), the pivot kernel starts logging the row factors of its etas, and the factorisation kernels run on a second stream into
// the OTHER set of factor arrays, on other compute units than the one-workgroup pivot kernel.
void Solver::start_async_refactor(long long iters_now) {
    const int m = d_.m;
    LuFactors& next = lu_sets_[lu_cur_ ^ 1];
    if (!refactor_stream_) {
        RELP_HIP(hipStreamCreateWithFlags(&refactor_stream_, hipStreamNonBlocking));
        RELP_HIP(hipEventCreateWithFlags(&ev_snapshot_, hipEventDisableTiming));
        RELP_HIP(hipEventCreateWithFlags(&ev_refactored_, hipEventDisableTiming));
        RELP_HIP(hipMalloc(&d_basis_snapshot_, (size_t)m * sizeof(int)));
        RELP_HIP(hipMalloc(&d_probe_, ((size_t)2 * m + 2) * sizeof(double)));  // the guard's probe v, B^-1 v and the residual
        launch_lu_probe_fill(d_probe_, m, stream_);
        if (bounded_) RELP_HIP(hipMalloc(&d_flipped_snapshot_, (size_t)d_.n * sizeof(*d_.flipped)));
    }
    if (!next.device_prepared()) next.prepare_device(m, refactor_period_ + 1, true, (size_t)h_col_start_.back());
    RELP_HIP(hipMemcpyAsync(d_basis_snapshot_, d_.basis, (size_t)m * sizeof(int), hipMemcpyDeviceToDevice, stream_));
    if (bounded_) RELP_HIP(hipMemcpyAsync(d_flipped_snapshot_, d_.flipped, (size_t)d_.n * sizeof(*d_.flipped), hipMemcpyDeviceToDevice, stream_));
    lu().start_log(stream_);
    RELP_HIP(hipEventRecord(ev_snapshot_, stream_));
    RELP_HIP(hipStreamWaitEvent(refactor_stream_, ev_snapshot_, 0));
    LuFactorSource src;
    src.col_start = d_.col_start;
    src.row_index = d_.row_index;
    src.value = d_.value;
    src.basis = d_basis_snapshot_;
    src.flipped = bounded_ ? reinterpret_cast<decltype(src.flipped)>(d_flipped_snapshot_) : nullptr;
    const double threshold = opt_.lu_pivot_threshold > 0.0 ? opt_.lu_pivot_threshold : 0.1;
    const int dense_tail = opt_.luf_dense_tail > 0 ? opt_.luf_dense_tail : (opt_.luf_dense_tail < 0 ? 0 : 8);
    next.refactor_device(src, threshold, 0, dense_tail, nullptr, ST_REFACTOR_FAILED, refactor_stream_);  // (a failure stays in its info words)
    RELP_HIP(hipEventRecord(ev_refactored_, refactor_stream_));
    async_in_flight_ = true;
    async_iters_at_snapshot_ = iters_now;
    if (diagnostic("RELP_TIME_REFACTOR")) fprintf(stderr, "[async] snapshot at iteration %lld, %lld updates since the last factors, set %d -> %d\n", iters_now, since_polish_, lu_cur_, lu_cur_ ^ 1);
}
// The new factors are ready (or are waited for): replay the logged etas onto them and swap sets.  false: they cannot be used (the
// kernels gave up, or more pivots were made than the log holds) -- the handle stays on the old factors and the caller takes the
// synchronous path when the device asks for a refactorisation.
bool Solver::finish_async_refactor(long long iters_now) {
    const double t0 = now_seconds();
    async_in_flight_ = false;
    LuFactors& next = lu_sets_[lu_cur_ ^ 1];
    RELP_HIP(hipStreamWaitEvent(stream_, ev_refactored_, 0));
    int info[LUF_INFO_WORDS];
    int state[LU_STATE_WORDS];
    RELP_HIP(hipMemcpyAsync(info, next.device_info(), sizeof(info), hipMemcpyDeviceToHost, stream_));
    RELP_HIP(hipMemcpyAsync(state, lu().device().state, sizeof(state), hipMemcpyDeviceToHost, stream_));
    RELP_HIP(hipStreamSynchronize(stream_));
    const long long made = iters_now - async_iters_at_snapshot_;
    if (diagnostic("RELP_TIME_REFACTOR")) {
        fprintf(stderr, "[async] ready: status %d, pivots since the snapshot %lld, logged %d, updates on the old factors %d (kept columns %d), iters %lld\n", info[LUF_STATUS], made,
                state[LU_LOG_COUNT], state[LU_N_UPDATES], state[LU_PF_COUNT], iters_now);
    }
    // (every basis change since the snapshot must be in the log: the count the kernels kept says so, the iteration counter is a
    //  cross-check -- bound flips make iterations without basis changes)
    if (info[LUF_STATUS] != LUF_OK || state[LU_LOG_COUNT] > LU_LOG_CAPACITY || state[LU_LOG_COUNT] > made) {
        ++async_abandoned_;
        if (info[LUF_STATUS] != LUF_OK) ++device_refactor_failures_;
        return false;
    }
    next.replay_log_of(lu(), stream_);
    // the guard (see lu_basis_residual_kernel): x = B^-1 v through the new factors and the replayed etas, |B x - v| over the basis now
    launch_lu_ftran_dense(next.device(), d_probe_, d_probe_ + d_.m, stream_);
    launch_lu_basis_residual(d_.col_start, d_.row_index, d_.value, d_.basis, bounded_ ? d_.flipped : nullptr, d_probe_ + d_.m, d_probe_, d_.m, d_probe_ + 2 * (size_t)d_.m, stream_);
    double residual = 0.0;
    RELP_HIP(hipMemcpyAsync(&residual, d_probe_ + 2 * (size_t)d_.m, sizeof(double), hipMemcpyDeviceToHost, stream_));
    RELP_HIP(hipStreamSynchronize(stream_));
    async_worst_residual_ = std::max(async_worst_residual_, residual);
    if (diagnostic("RELP_TIME_REFACTOR")) fprintf(stderr, "[async] |B (B^-1 v) - v| through the new factors + %d replayed etas: %.3e\n", state[LU_LOG_COUNT], residual);
    if (!(residual <= 1e-8)) {  // (the chain of swaps has drifted: this cycle ends with a factorisation of the basis as it is)
        ++async_abandoned_;
        return false;
    }
    lu_cur_ ^= 1;
    launch_lu_xb(d_, lu().device(), stream_);
    launch_lu_pi(d_, lu().device(), stream_);
    binv_identity_ = false;
    refactors_++;
    async_refactors_++;
    since_polish_ = state[LU_LOG_COUNT];
    refactor_seconds_ += now_seconds() - t0;
    return true;
}
void Solver::refactor_lu_host(bool refresh_vectors) {
    abandon_async_flight();
    const double t0 = now_seconds();
    const int m = d_.m;
    std::vector<int> basis(m);
    RELP_HIP(hipMemcpyAsync(basis.data(), d_.basis, m * sizeof(int), hipMemcpyDeviceToHost, stream_));
    RELP_HIP(hipStreamSynchronize(stream_));
    std::vector<int> cs(m + 1, 0);
    size_t total = 0;
    for (int k = 0; k < m; ++k) {
        if (basis[k] < 0 || basis[k] >= d_.n) throw std::runtime_error("the device returned an invalid basis");
        total += (size_t)(h_col_start_[basis[k] + 1] - h_col_start_[basis[k]]);
        cs[k + 1] = (int)total;
    }
    std::vector<int> rows(total);
    std::vector<double> vals(total);
    std::vector<int> flipped;
    if (bounded_) {  // implicit bounds: a complemented column sits in the basis with the opposite sign
        flipped.resize(d_.n);
        RELP_HIP(hipMemcpyAsync(flipped.data(), d_.flipped, d_.n * sizeof(int), hipMemcpyDeviceToHost, stream_));
        RELP_HIP(hipStreamSynchronize(stream_));
    }
    for (int k = 0; k < m; ++k) {
        const int a = h_col_start_[basis[k]], len = h_col_start_[basis[k] + 1] - a;
        std::copy(h_row_index_.begin() + a, h_row_index_.begin() + a + len, rows.begin() + cs[k]);
        std::copy(h_value_.begin() + a, h_value_.begin() + a + len, vals.begin() + cs[k]);
        if (bounded_ && flipped[basis[k]])
            for (int e = cs[k]; e < cs[k] + len; ++e) vals[e] = -vals[e];
    }
    LuOptions lo;
    lo.threshold = opt_.lu_pivot_threshold > 0.0 ? opt_.lu_pivot_threshold : 0.1;
    static const bool time_parts = diagnostic("RELP_TIME_REFACTOR");
    thread_local double part_seconds[3] = {0.0, 0.0, 0.0};
    const double t1 = now_seconds();
    HostLU f = lu_factor(m, cs.data(), rows.data(), vals.data(), lo);
    if (f.singular) throw std::runtime_error("singular basis in the LU refactorisation");
    const double t2 = now_seconds();
    if (lu().upload(f, refactor_period_ + 1, stream_, lu_inverse_)) destroy_graphs();
    if (time_parts) {
        part_seconds[0] += t1 - t0;
        part_seconds[1] += t2 - t1;
        part_seconds[2] += now_seconds() - t2;
        if ((refactors_ + 1) % 25 == 0)
            fprintf(stderr, "[refactor] %lld so far: basis fetch + gather %.2f ms, Markowitz LU %.2f ms, schedules + upload %.2f ms (sums)\n", refactors_ + 1,
                    part_seconds[0] * 1e3, part_seconds[1] * 1e3, part_seconds[2] * 1e3);
    }
    binv_identity_ = false;
    if (refresh_vectors) {
        launch_lu_xb(d_, lu().device(), stream_);
        launch_lu_pi(d_, lu().device(), stream_);  // also rewrites minus_obj from the refreshed x_B
    }
    // status: REFACTOR -> RUNNING, stream-ordered (no host round trip: the refresh kernels above are still running; whoever reads
    // the control block next synchronises anyway).  A refactorisation is only ever asked for in the RUNNING state.
    launch_clear_refactor_status(d_, stream_);
    refactors_++;
    since_polish_ = 0;
    refactor_seconds_ += now_seconds() - t0;
    if (opt_.verbose > 1)
        fprintf(stderr, "[lu] refactor %lld: nnz(L) %lld nnz(U) %lld depth %d + %d\n", refactors_, lu().nnz_l, lu().nnz_u, lu().depth_l, lu().depth_u);
}

// ---- fine-grained ops -------------------------------------------------------------------------------
void Solver::ftran(int nnz, const int* rows, const double* values, double* out) {
    int* d_rows = reinterpret_cast<int*>(d_.scratch);
    double* d_vals = d_.scratch + (d_.m + 1) / 2 + 1;
    double* d_out = d_vals + d_.m;
    check_sparse(nnz, rows, values, d_.m);
    RELP_HIP(hipSetDevice(opt_.device));
    RELP_HIP(hipMemcpyAsync(d_rows, rows, nnz * sizeof(int), hipMemcpyHostToDevice, stream_));
    RELP_HIP(hipMemcpyAsync(d_vals, values, nnz * sizeof(double), hipMemcpyHostToDevice, stream_));
    if (lu_mode_) launch_lu_ftran(lu().device(), d_rows, d_vals, nnz, d_out, 0, stream_);
    else launch_ftran_vec(d_, d_rows, d_vals, nnz, d_out, stream_);
    RELP_HIP(hipMemcpyAsync(out, d_out, d_.m * sizeof(double), hipMemcpyDeviceToHost, stream_));
    RELP_HIP(hipStreamSynchronize(stream_));
}
void Solver::btran(int nnz, const int* rows, const double* values, double* out) {
    int* d_rows = reinterpret_cast<int*>(d_.scratch);
    double* d_vals = d_.scratch + (d_.m + 1) / 2 + 1;
    double* d_out = d_vals + d_.m;
    check_sparse(nnz, rows, values, d_.m);
    RELP_HIP(hipSetDevice(opt_.device));
    RELP_HIP(hipMemcpyAsync(d_rows, rows, nnz * sizeof(int), hipMemcpyHostToDevice, stream_));
    RELP_HIP(hipMemcpyAsync(d_vals, values, nnz * sizeof(double), hipMemcpyHostToDevice, stream_));
    if (lu_mode_) launch_lu_btran(lu().device(), d_rows, d_vals, nnz, d_out, stream_);
    else launch_btran_vec(d_, d_rows, d_vals, nnz, d_out, stream_);
    RELP_HIP(hipMemcpyAsync(out, d_out, d_.m * sizeof(double), hipMemcpyDeviceToHost, stream_));
    RELP_HIP(hipStreamSynchronize(stream_));
}
void Solver::inverse_row(int row, double* out) {
    if (row < 0 || row >= d_.m) throw std::invalid_argument("row out of range");
    const double one = 1.0;
    btran(1, &row, &one, out);  // e_row' Binv: a strided read of the column-major inverse
}
void Solver::relative_costs(double* out) {
    if (phase_ == 0) throw std::runtime_error("no phase started");
    RELP_HIP(hipSetDevice(opt_.device));
    launch_relative_cost(d_, d_.scratch, stream_);
    RELP_HIP(hipMemcpyAsync(out, d_.scratch, d_.n * sizeof(double), hipMemcpyDeviceToHost, stream_));
    RELP_HIP(hipStreamSynchronize(stream_));
}
void Solver::get_gamma(double* out) {
    RELP_HIP(hipSetDevice(opt_.device));
    // apply a pending weight update first so that the values are those the next pricing pass would use
    std::vector<int> pos(d_.n);
    RELP_HIP(hipMemcpyAsync(out, d_.gamma, d_.n * sizeof(double), hipMemcpyDeviceToHost, stream_));
    RELP_HIP(hipMemcpyAsync(pos.data(), d_.pos, d_.n * sizeof(int), hipMemcpyDeviceToHost, stream_));
    RELP_HIP(hipStreamSynchronize(stream_));
    for (int j = 0; j < d_.n; ++j)
        if (j < d_.n_art || pos[j] >= 0) out[j] = std::numeric_limits<double>::quiet_NaN();
}
// `PivotRule::select_primal_pivot_column`: the pricing kernel, then the entering-column reduction of the fused kernel
// (mode 1: stop after the choice).  Applies a pending steepest-edge update exactly like the device loop does.
void Solver::price(int* column, double* cbar) {
    if (phase_ == 0) throw std::runtime_error("no phase started");
    RELP_HIP(hipSetDevice(opt_.device));
    Ctl c = read_ctl();
    const int saved = c.status;
    c.status = ST_RUNNING;
    c.forced_q = c.forced_p = -1;
    write_ctl(c);
    enqueue_price(0);
    if (lu_mode_) enqueue_ftran_ratio(1);
    else launch_ftran_ratio(d_, opt_.pivot_rule, price_blocks_ + dense_blocks_, opt_.tol_pivot, ratio_delta(), phase_ == 2 ? 1 : 0, 1, 0, stream_);
    c = read_ctl();
    *column = c.q;
    *cbar = c.q >= 0 ? c.cbar_q : 0.0;
    c.status = saved;
    write_ctl(c);
}
// `Tableau::generate_column` + `select_primal_pivot_row` without a basis change: the fused kernel in mode 2.
void Solver::ratio(int column, int* row, double* alpha_out) {
    if (phase_ == 0) throw std::runtime_error("no phase started");
    RELP_HIP(hipSetDevice(opt_.device));
    if (column < 0 || column >= d_.n) throw std::invalid_argument("column out of range");
    Ctl c = read_ctl();
    const Ctl before = c;  // a pending steepest-edge update still needs q, gamma_q, alpha_pq, leaving of the LAST pivot
    const int saved = c.status;
    const int saved_pending = c.pending;
    c.status = ST_RUNNING;
    c.forced_q = column;
    c.forced_p = -1;
    write_ctl(c);
    enqueue_ftran_ratio(2);
    c = read_ctl();
    *row = c.p;
    if (alpha_out) {
        RELP_HIP(hipMemcpyAsync(alpha_out, d_.alpha, d_.m * sizeof(double), hipMemcpyDeviceToHost, stream_));
        RELP_HIP(hipStreamSynchronize(stream_));
    }
    c.status = saved;
    c.pending = saved_pending;
    c.q = before.q;
    c.p = before.p;
    c.leaving = before.leaving;
    c.cbar_q = before.cbar_q;
    c.alpha_pq = before.alpha_pq;
    c.gamma_q = before.gamma_q;
    c.forced_q = c.forced_p = -1;
    write_ctl(c);
}
void exact_simplex(const StandardForm& form, int device, hipStream_t stream, int first_limbs, int max_limbs, long long max_pivots,
                   int trace_capacity, int* status, int* limbs_used, long long* pivots_phase_one, long long* pivots_phase_two,
                   std::vector<int>* trace, std::string* objective, std::vector<int>* final_basis,
                   std::vector<std::pair<int, long long>>* pivots_survived, int* redundant_rows, std::vector<ExactWidthRecord>* counters, int update_mode, int forced_grid);
void Solver::solve_exact(int first_limbs, int max_limbs, long long max_pivots, int trace_capacity, int* status, int* limbs, long long* p1,
                         long long* p2, std::vector<int>* trace, std::string* objective, std::vector<int>* basis,
                         std::vector<std::pair<int, long long>>* survived, int* redundant_rows) {
    if (!loaded_) throw std::runtime_error("no LP loaded");
    const long long cap = max_pivots > 0 ? max_pivots : 200LL * (d_.m + d_.n) + 100000;
    exact_simplex(form_, opt_.device, stream_, first_limbs, max_limbs, cap, trace_capacity, status, limbs, p1, p2, trace, objective, basis, survived,
                  redundant_rows, &exact_records_, opt_.exact_update, opt_.exact_grid);
}
void Solver::last_pivot(int* phase, int* column, int* row, int* leaving) {
    const Ctl c = read_ctl();
    const bool any = c.iters > 0 && c.p >= 0;
    *phase = phase_;
    *column = any ? c.q : -1;
    *row = any ? c.p : -1;
    *leaving = any ? c.leaving : -1;
}
// `PivotRule::after_basis_update` (strategy/pivot_rule.rs:243-296): the Goldfarb-Reid update of the steepest-edge weights for
// the last basis change.  Inside the device loop it rides on the next pricing pass; a caller that drives the loop itself can
// ask for it here (one pass over the non-basic columns, candidates discarded).  No-op when nothing is pending.
void Solver::after_basis_update() {
    if (phase_ == 0) throw std::runtime_error("no phase started");
    RELP_HIP(hipSetDevice(opt_.device));
    Ctl c = read_ctl();
    if (!c.pending) return;
    const int saved = c.status;
    c.status = ST_RUNNING;
    write_ctl(c);
    enqueue_price(0);
    c = read_ctl();
    c.pending = 0;
    c.status = saved;
    write_ctl(c);
}
// `Tableau::bring_into_basis(pivot_column, pivot_row, column)` (tableau/mod.rs:139-160) with both indices given by the
// caller: one iteration of the device loop whose pricing and ratio test are overridden (the steepest-edge weights, b,
// -pi, the objective and the inverse are updated as in any other pivot).  Same index space as `price` / `ratio`.
void Solver::bring_into_basis(int column, int row) {
    if (phase_ == 0) throw std::runtime_error("no phase started");
    if (column < 0 || column >= d_.n) throw std::invalid_argument("column out of range");
    if (row < 0 || row >= d_.m) throw std::invalid_argument("row out of range");
    Ctl c = read_ctl();
    const long long before = c.iters;
    c.status = ST_RUNNING;
    c.forced_q = column;
    c.forced_p = row;
    write_ctl(c);
    launch_pivots(1, true);
    c = read_ctl();
    if (c.iters == before) throw std::runtime_error("bring_into_basis: the pivot element is zero (or the column is basic)");
    pivots_[phase_ - 1] += 1;
    since_polish_ += 1;
    binv_identity_ = false;
    if (c.status == ST_REFACTOR) refactor_lu(true);  // LU carry: `should_refactor` was true for this pivot
}
// `BasisInverse::should_refactor` + `invert` (lower_upper/mod.rs:78-92, 249-252) on demand: the Newton-Schulz polish of the
// resident inverse, with b, -pi and the objective recomputed from it.  Returns the residual max|I - B'T| found before.
double Solver::refactor() {
    if (phase_ == 0) throw std::runtime_error("no phase started");
    const double before = max_residual_;
    max_residual_ = 0.0;
    polish(true, true);
    const double found = max_residual_;
    max_residual_ = std::max(before, found);
    return found;
}
// Average execution time of one launch of a hot-loop kernel INSIDE the real pivot sequence (bench.py's roofline leg):
// `repetitions` further pivots of the current phase are run un-graphed, and the chosen kernel of every pivot is
// bracketed by its own start/stop event pair (hipExtLaunchKernelGGL) on this handle's stream.  The solve advances.
double Solver::profile_kernel(int which, int repetitions) {
    if (phase_ == 0) throw std::runtime_error("no phase started");
    if (lu_mode_ && which == 2) throw std::invalid_argument("the LU carry has no separate update kernel (which = 1 covers it)");
    if (fused_ && which == 2) throw std::invalid_argument("the update is part of kernel 1 (fused pivot kernel): which = 1 covers it");
    RELP_HIP(hipSetDevice(opt_.device));
    const int m = d_.m;
    if (which == 0) {
        // algorithmic bytes of a pricing pass at this state: the columns that are non-basic now
        std::vector<int> pos(d_.n), cs(d_.n + 1);
        RELP_HIP(hipMemcpyAsync(pos.data(), d_.pos, d_.n * sizeof(int), hipMemcpyDeviceToHost, stream_));
        RELP_HIP(hipMemcpyAsync(cs.data(), d_.col_start, (d_.n + 1) * sizeof(int), hipMemcpyDeviceToHost, stream_));
        RELP_HIP(hipStreamSynchronize(stream_));
        long long bytes = 0;
        for (int j = d_.n_art; j < d_.n; ++j) {
            if (pos[j] >= 0) continue;
            if (d_.cost8) {  // generated incidence column: 8 B of endpoints, 1 B cost, 4 B position
                bytes += 13;
                continue;
            }
            const bool dense_col = j >= d_.dense_first && j < d_.dense_first + d_.n_dense;
            bytes += dense_col ? (long long)m * dense_entry_bytes_ : (long long)(cs[j + 1] - cs[j]) * 12;
            bytes += 24;  // cost, gamma read + gamma write
        }
        stats_.price_bytes = bytes;
    }
    std::vector<hipEvent_t> starts(repetitions), stops(repetitions);
    for (int k = 0; k < repetitions; ++k) {
        RELP_HIP(hipEventCreate(&starts[k]));
        RELP_HIP(hipEventCreate(&stops[k]));
    }
    Ctl before = read_ctl();
    if (fused_) {
        launch_begin_batch(d_, repetitions, stream_);
        for (int k = 0; k < repetitions; ++k) {
            if (which == 0) arm_launch_timer(0, starts[k], stops[k]);
            enqueue_price_fused(k & 1);
            if (which == 1) arm_launch_timer(1, starts[k], stops[k]);
            enqueue_pivot_fused(k & 1);
        }
        launch_commit(d_, repetitions & 1, stream_);
    } else {
    launch_budget(d_, repetitions, stream_);
    for (int k = 0; k < repetitions; ++k) {
        if (which == 0) arm_launch_timer(0, starts[k], stops[k]);
        enqueue_price(0, k == 0);
        if (which == 1) arm_launch_timer(1, starts[k], stops[k]);
        enqueue_ftran_ratio(0);
        if (which == 2) arm_launch_timer(2, starts[k], stops[k]);
        if (!lu_mode_) enqueue_update();
        if (eta_mode_ && ((k + 1) % d_.eta_cap == 0 || k + 1 == repetitions)) enqueue_consolidate();
    }
    }
    arm_launch_timer(-1, nullptr, nullptr);
    Ctl after = read_ctl();
    // K3's algorithmic bytes at the profiled state when unit columns are skipped: read + write of (non-zero rows of alpha) x
    // (columns of the stored inverse that carry information)
    if (d_.track_touched && !eta_mode_) stats_.update_bytes = 16LL * std::max(1, after.nz_count) * std::max(1, after.touched_count);
    if (after.status == ST_REFACTOR) refactor_lu(true);
    const long long made = after.iters - before.iters;
    pivots_[phase_ - 1] += made;
    since_polish_ += made;
    double total_ms = 0.0;
    int counted = 0;
    for (int k = 0; k < repetitions; ++k) {
        float ms = 0.f;
        if (k < made && hipEventElapsedTime(&ms, starts[k], stops[k]) == hipSuccess) {
            total_ms += ms;
            ++counted;
        }
        (void)hipEventDestroy(starts[k]);
        (void)hipEventDestroy(stops[k]);
    }
    if (counted == 0) throw std::runtime_error("no pivot was made while profiling (phase already finished)");
    return total_ms * 1e-3 / counted;
}

void Solver::debug_stamps(unsigned long long* out64) {
    RELP_HIP(hipMemcpyAsync(out64, d_.dbg, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream_));
    RELP_HIP(hipStreamSynchronize(stream_));
}

void Solver::get_b(double* out) {
    RELP_HIP(hipSetDevice(opt_.device));
    RELP_HIP(hipMemcpyAsync(out, d_.xB, d_.m * sizeof(double), hipMemcpyDeviceToHost, stream_));
    RELP_HIP(hipStreamSynchronize(stream_));
}
double Solver::objective() { return -read_ctl().minus_obj; }
void Solver::get_basis(int* out) {
    std::vector<int> basis(d_.m);
    RELP_HIP(hipMemcpyAsync(basis.data(), d_.basis, d_.m * sizeof(int), hipMemcpyDeviceToHost, stream_));
    RELP_HIP(hipStreamSynchronize(stream_));
    if (bounded_) {  // in the reference's formulation: one entry per row of MatrixData, bound rows included
        std::vector<int> pos(d_.n);
        RELP_HIP(hipMemcpyAsync(pos.data(), d_.pos, d_.n * sizeof(int), hipMemcpyDeviceToHost, stream_));
        RELP_HIP(hipStreamSynchronize(stream_));
        resolve_fixed_columns(pos);
        const std::vector<int> full = explicit_basis(basis, pos);
        std::copy(full.begin(), full.end(), out);
        return;
    }
    for (int i = 0; i < d_.m; ++i) out[i] = basis[i] >= d_.n_art ? basis[i] - d_.n_art : -1 - basis[i];
}
void Solver::get_solution(double* x) const {
    const int n_struct = form_.data.nr_normal_variables();
    for (int j = 0; j < n_struct; ++j) x[j] = h_solution_[j];
}

}  // namespace relp

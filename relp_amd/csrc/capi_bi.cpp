// extern "C" boundary of the stand-alone `BasisInverse` (include/relp_amd.h, section "BasisInverse as an object").
#include <cstring>
#include <memory>
#include <string>

#include "lu.hpp"
#include "lu_factor.hpp"
#include "solver.hpp"

using namespace relp;

struct relp_basis_inverse {
    std::unique_ptr<LuBasis> lu;
    std::string error;
    unsigned switches = 0;  // relp_bi_options.switches: in force on the calling thread during every call on this object
};

namespace {
thread_local std::string g_bi_error;

template <class F>
int32_t guarded_bi(relp_basis_inverse* h, F&& f) {
    auto note = [&](const char* what) {
        g_bi_error = what;
        if (h) h->error = what;
    };
    try {
        Tuning tuning = thread_tuning();
        if (h) tuning.switches |= h->switches;
        const TuningScope scope(tuning);
        f();
        return RELP_OK;
    } catch (const DeviceError& e) {
        note(e.what());
        return RELP_ERR_DEVICE;
    } catch (const std::invalid_argument& e) {
        note(e.what());
        return RELP_ERR_ARGUMENT;
    } catch (const std::logic_error& e) {
        note(e.what());
        return RELP_ERR_STATE;
    } catch (const std::bad_alloc& e) {
        note(e.what());
        return RELP_ERR_STATE;
    } catch (const std::exception& e) {
        note(e.what());
        return RELP_ERR_NUMERICAL;
    }
}
LuOptions lu_options_of(const relp_bi_options* o) {
    LuOptions lo;
    if (o) {
        lo.threshold = o->pivot_threshold;
        lo.reference_ties = o->reference_ties != 0;
    }
    return lo;
}
template <class T, class U>
bool copy_out(const std::vector<T>& src, U* dst, int64_t capacity) {
    if ((int64_t)src.size() > capacity) return false;
    if (dst)
        for (size_t k = 0; k < src.size(); ++k) dst[k] = (U)src[k];
    return true;
}
}  // namespace

extern "C" {

int32_t relp_bi_options_default(relp_bi_options* o) {
    if (!o) return RELP_ERR_ARGUMENT;
    std::memset(o, 0, sizeof(*o));
    o->device = 0;
    o->refactor_period = 31;   // lower_upper/mod.rs:249-252
    o->pivot_threshold = 0.1;
    o->reference_ties = 0;
    return RELP_OK;
}
const char* relp_bi_last_error(const relp_basis_inverse* bi) { return bi ? bi->error.c_str() : g_bi_error.c_str(); }

int32_t relp_bi_identity(const relp_bi_options* options, int32_t m, relp_basis_inverse** out) {
    if (!out) return RELP_ERR_ARGUMENT;
    *out = nullptr;
    std::unique_ptr<relp_basis_inverse> h(new relp_basis_inverse());
    h->switches = options ? (unsigned)options->switches : 0u;
    const int32_t status = guarded_bi(h.get(), [&] {
        h->lu.reset(new LuBasis(options ? options->device : 0, m, lu_options_of(options), options ? options->refactor_period : 31));
        h->lu->identity();
    });
    if (status == RELP_OK) *out = h.release();
    return status;
}
int32_t relp_bi_invert(const relp_bi_options* options, int32_t m, const int64_t* column_start, const int32_t* row_index,
                       const double* value, relp_basis_inverse** out) {
    if (!out || !column_start || (column_start[m > 0 ? m : 0] > 0 && (!row_index || !value))) return RELP_ERR_ARGUMENT;
    *out = nullptr;
    std::unique_ptr<relp_basis_inverse> h(new relp_basis_inverse());
    h->switches = options ? (unsigned)options->switches : 0u;
    const int32_t status = guarded_bi(h.get(), [&] {
        h->lu.reset(new LuBasis(options ? options->device : 0, m, lu_options_of(options), options ? options->refactor_period : 31));
        static_assert(sizeof(long long) == sizeof(int64_t), "");
        h->lu->invert(reinterpret_cast<const long long*>(column_start), row_index, value);
    });
    if (status == RELP_OK) *out = h.release();
    return status;
}
int32_t relp_bi_free(relp_basis_inverse* bi) {
    delete bi;
    return RELP_OK;
}
int32_t relp_bi_m(const relp_basis_inverse* bi, int32_t* m) {
    if (!bi || !m) return RELP_ERR_ARGUMENT;
    *m = bi->lu->m();
    return RELP_OK;
}
int32_t relp_bi_left_multiply(relp_basis_inverse* bi, int32_t nnz, const int32_t* row_index, const double* value, double* out_m) {
    if (!bi || !out_m) return RELP_ERR_ARGUMENT;
    return guarded_bi(bi, [&] { bi->lu->left_multiply(nnz, row_index, value, out_m); });
}
int32_t relp_bi_right_multiply(relp_basis_inverse* bi, int32_t nnz, const int32_t* index, const double* value, double* out_m) {
    if (!bi || !out_m) return RELP_ERR_ARGUMENT;
    return guarded_bi(bi, [&] { bi->lu->right_multiply(nnz, index, value, out_m); });
}
int32_t relp_bi_basis_inverse_row(relp_basis_inverse* bi, int32_t row, double* out_m) {
    if (!bi || !out_m) return RELP_ERR_ARGUMENT;
    return guarded_bi(bi, [&] {
        if (row < 0 || row >= bi->lu->m()) throw std::invalid_argument("row out of range");
        bi->lu->basis_inverse_row(row, out_m);
    });
}
int32_t relp_bi_generate_element(relp_basis_inverse* bi, int32_t i, int32_t nnz, const int32_t* row_index, const double* value,
                                 double* element, int32_t* is_some) {
    if (!bi || !element) return RELP_ERR_ARGUMENT;
    return guarded_bi(bi, [&] {
        const bool some = bi->lu->generate_element(i, nnz, row_index, value, element);
        if (is_some) *is_some = some ? 1 : 0;
    });
}
int32_t relp_bi_change_basis(relp_basis_inverse* bi, int32_t pivot_row_index) {
    if (!bi) return RELP_ERR_ARGUMENT;
    return guarded_bi(bi, [&] { bi->lu->change_basis(pivot_row_index); });
}
int32_t relp_bi_should_refactor(relp_basis_inverse* bi, int32_t* should) {
    if (!bi || !should) return RELP_ERR_ARGUMENT;
    return guarded_bi(bi, [&] { *should = bi->lu->should_refactor() ? 1 : 0; });
}
int32_t relp_bi_remove_basis_part(relp_basis_inverse* bi, int32_t count, const int32_t* indices) {
    if (!bi || (count > 0 && !indices) || count < 0) return RELP_ERR_ARGUMENT;
    return guarded_bi(bi, [&] { bi->lu->remove_basis_part(count, indices); });
}
int32_t relp_bi_statistics(relp_basis_inverse* bi, int64_t* nnz_lower, int64_t* nnz_upper, int32_t* depth_lower,
                           int32_t* depth_upper, int32_t* nr_updates) {
    if (!bi) return RELP_ERR_ARGUMENT;
    return guarded_bi(bi, [&] {
        if (nnz_lower) *nnz_lower = bi->lu->nnz_l();
        if (nnz_upper) *nnz_upper = bi->lu->nnz_u();
        if (depth_lower) *depth_lower = bi->lu->depth_l();
        if (depth_upper) *depth_upper = bi->lu->depth_u();
        if (nr_updates) *nr_updates = bi->lu->updates();
    });
}

int32_t relp_bi_get_factors(relp_basis_inverse* bi, int64_t capacity, int32_t* row_permutation, int32_t* column_permutation,
                            int64_t* lower_start, int32_t* lower_row, double* lower_value, int64_t* upper_start,
                            int32_t* upper_row, double* upper_value, double* upper_diagonal, int32_t* nr_updates,
                            int64_t* eta_start, int32_t* eta_pivot, int32_t* eta_index, double* eta_value) {
    if (!bi) return RELP_ERR_ARGUMENT;
    return guarded_bi(bi, [&] {
        const LuBasis::Factors f = bi->lu->factors();
        const bool ok = copy_out(f.row_permutation, row_permutation, capacity) && copy_out(f.column_permutation, column_permutation, capacity) &&
                        copy_out(f.l_start, lower_start, capacity) && copy_out(f.l_row, lower_row, capacity) && copy_out(f.l_val, lower_value, capacity) &&
                        copy_out(f.u_start, upper_start, capacity) && copy_out(f.u_row, upper_row, capacity) && copy_out(f.u_val, upper_value, capacity) &&
                        copy_out(f.upper_diagonal, upper_diagonal, capacity) && copy_out(f.eta_start, eta_start, capacity) &&
                        copy_out(f.eta_pivot, eta_pivot, capacity) && copy_out(f.eta_index, eta_index, capacity) &&
                        copy_out(f.eta_value, eta_value, capacity);
        if (!ok) throw std::invalid_argument("capacity too small");
        if (nr_updates) *nr_updates = (int32_t)f.eta_pivot.size();
    });
}

// ---- host only: the factorisation step alone (no device) ----------------------------------------------------------------
int32_t relp_lu_factor_host(int32_t m, const int64_t* column_start, const int32_t* row_index, const double* value,
                            double pivot_threshold, int32_t reference_ties, int64_t capacity, int32_t* row_permutation,
                            int32_t* column_permutation, int64_t* lower_start, int32_t* lower_column, double* lower_value,
                            int64_t* upper_start, int32_t* upper_column, double* upper_value, double* upper_diagonal,
                            int32_t* depth_lower, int32_t* depth_upper) {
    if (m < 1 || !column_start) return RELP_ERR_ARGUMENT;
    return guarded_bi(nullptr, [&] {
        std::vector<int> cs(m + 1);
        for (int j = 0; j <= m; ++j) cs[j] = (int)column_start[j];
        for (int64_t e = 0; e < column_start[m]; ++e)
            if (row_index[e] < 0 || row_index[e] >= m) throw std::invalid_argument("row index out of range");
        LuOptions lo;
        lo.threshold = pivot_threshold;
        lo.reference_ties = reference_ties != 0;
        const HostLU f = lu_factor(m, cs.data(), row_index, value, lo);
        if (f.singular) throw std::runtime_error("singular basis");
        const bool ok = copy_out(f.rowpos, row_permutation, capacity) && copy_out(f.colpos, column_permutation, capacity) &&
                        copy_out(f.l_start, lower_start, capacity) && copy_out(f.l_col, lower_column, capacity) && copy_out(f.l_val, lower_value, capacity) &&
                        copy_out(f.u_start, upper_start, capacity) && copy_out(f.u_col, upper_column, capacity) && copy_out(f.u_val, upper_value, capacity) &&
                        copy_out(f.diag, upper_diagonal, capacity);
        if (!ok) throw std::invalid_argument("capacity too small");
        int dl = 0, du = 0;
        lu_depths(f, &dl, &du);
        if (depth_lower) *depth_lower = dl;
        if (depth_upper) *depth_upper = du;
    });
}

// ... and the same factorisation with both triangles INVERTED as sparse matrices, what the inverse-factor carry uploads (lu.hpp;
// `lu_invert_factors`): lower_* = the strict part of L^-1 by rows (unit diagonal implied), upper_* = U^-1 by rows WITH its diagonal,
// upper_diagonal = ones.  Host only, for tests.
int32_t relp_lu_invert_host(int32_t m, const int64_t* column_start, const int32_t* row_index, const double* value,
                            double pivot_threshold, int64_t capacity, int32_t* row_permutation, int32_t* column_permutation,
                            int64_t* lower_start, int32_t* lower_column, double* lower_value, int64_t* upper_start,
                            int32_t* upper_column, double* upper_value, double* upper_diagonal) {
    if (m < 1 || !column_start) return RELP_ERR_ARGUMENT;
    return guarded_bi(nullptr, [&] {
        std::vector<int> cs(m + 1);
        for (int j = 0; j <= m; ++j) cs[j] = (int)column_start[j];
        for (int64_t e = 0; e < column_start[m]; ++e)
            if (row_index[e] < 0 || row_index[e] >= m) throw std::invalid_argument("row index out of range");
        LuOptions lo;
        lo.threshold = pivot_threshold;
        const HostLU f = lu_factor(m, cs.data(), row_index, value, lo);
        if (f.singular) throw std::runtime_error("singular basis");
        HostLU inverted;
        if (!lu_invert_factors(f, (size_t)capacity, inverted)) throw std::invalid_argument("capacity too small");
        const bool ok = copy_out(inverted.rowpos, row_permutation, capacity) && copy_out(inverted.colpos, column_permutation, capacity) &&
                        copy_out(inverted.l_start, lower_start, capacity) && copy_out(inverted.l_col, lower_column, capacity) &&
                        copy_out(inverted.l_val, lower_value, capacity) && copy_out(inverted.u_start, upper_start, capacity) &&
                        copy_out(inverted.u_col, upper_column, capacity) && copy_out(inverted.u_val, upper_value, capacity) &&
                        copy_out(inverted.diag, upper_diagonal, capacity);
        if (!ok) throw std::invalid_argument("capacity too small");
    });
}


// ---- the factorisation step as the device runs it (lu_factor.hip), for tests -------------------------------------------------
int32_t relp_lu_factor_device(int32_t device, int32_t m, const int64_t* column_start, const int32_t* row_index, const double* value,
                              double pivot_threshold, int32_t reference_ties, int32_t dense_tail, int32_t inverted, int64_t capacity,
                              int32_t* row_permutation, int32_t* column_permutation, int64_t* lower_start, int32_t* lower_column,
                              double* lower_value, int64_t* upper_start, int32_t* upper_column, double* upper_value,
                              double* upper_diagonal, int32_t* info) {
    if (m < 1 || !column_start || capacity < m + 1) return RELP_ERR_ARGUMENT;
    return guarded_bi(nullptr, [&] {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) throw DeviceError("no HIP device");
        RELP_HIP(hipSetDevice(device));
        const size_t nnz = (size_t)column_start[m];
        std::vector<int> cs(m + 1);
        for (int j = 0; j <= m; ++j) cs[j] = (int)column_start[j];
        for (size_t e = 0; e < nnz; ++e)
            if (row_index[e] < 0 || row_index[e] >= m) throw std::invalid_argument("row index out of range");
        const size_t cap = (size_t)capacity;
        struct Buffers {
            int *cs = nullptr, *ri = nullptr, *rowpos = nullptr, *colpos = nullptr, *ls = nullptr, *lc = nullptr, *us = nullptr, *uc = nullptr;
            double *va = nullptr, *diag = nullptr, *lv = nullptr, *uv = nullptr;
            ~Buffers() {
                for (void* p : {(void*)cs, (void*)ri, (void*)rowpos, (void*)colpos, (void*)ls, (void*)lc, (void*)us, (void*)uc, (void*)va, (void*)diag, (void*)lv, (void*)uv})
                    if (p) (void)hipFree(p);
            }
        } b;
        auto alloc_i = [&](int** p, size_t n) { RELP_HIP(hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(1, n) * sizeof(int))); };
        auto alloc_d = [&](double** p, size_t n) { RELP_HIP(hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(1, n) * sizeof(double))); };
        alloc_i(&b.cs, m + 1); alloc_i(&b.ri, nnz); alloc_d(&b.va, nnz);
        alloc_i(&b.rowpos, m); alloc_i(&b.colpos, m); alloc_d(&b.diag, m);
        alloc_i(&b.ls, m + 1); alloc_i(&b.lc, cap); alloc_d(&b.lv, cap);
        alloc_i(&b.us, m + 1); alloc_i(&b.uc, cap); alloc_d(&b.uv, cap);
        RELP_HIP(hipMemcpy(b.cs, cs.data(), (m + 1) * sizeof(int), hipMemcpyHostToDevice));
        if (nnz) {
            RELP_HIP(hipMemcpy(b.ri, row_index, nnz * sizeof(int), hipMemcpyHostToDevice));
            RELP_HIP(hipMemcpy(b.va, value, nnz * sizeof(double), hipMemcpyHostToDevice));
        }
        LuFactorScratch scratch;
        scratch.reserve(m, nnz, cap, cap, inverted ? cap : 0);
        LuFactorSource src;
        src.col_start = b.cs;
        src.row_index = b.ri;
        src.value = b.va;
        LuFactorOut out;
        out.rowpos = b.rowpos; out.colpos = b.colpos; out.diag = b.diag;
        out.l_start = b.ls; out.l_col = b.lc; out.l_val = b.lv;
        out.u_start = b.us; out.u_col = b.uc; out.u_val = b.uv;
        out.cap_l = out.cap_u = (int)std::min<size_t>(cap, (size_t)1 << 30);
        // twice: the second run (warm caches, the kernel's code resident) is timed with HIP events -> info[31] in units of 0.1 us
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        RELP_HIP(hipEventCreate(&ev0));
        RELP_HIP(hipEventCreate(&ev1));
        launch_lu_factor(src, scratch.work(), out, pivot_threshold, reference_ties, dense_tail, nullptr);
        RELP_HIP(hipEventRecord(ev0, nullptr));
        launch_lu_factor(src, scratch.work(), out, pivot_threshold, reference_ties, dense_tail, nullptr);
        if (inverted) launch_lu_invert(out, scratch.inverse_work(), scratch.work().info, nullptr);  // (timed with the factorisation)
        RELP_HIP(hipEventRecord(ev1, nullptr));
        RELP_HIP(hipDeviceSynchronize());
        float elapsed_ms = 0.0f;
        (void)hipEventElapsedTime(&elapsed_ms, ev0, ev1);
        (void)hipEventDestroy(ev0);
        (void)hipEventDestroy(ev1);
        std::vector<int> h_info(LUF_INFO_WORDS);
        RELP_HIP(hipMemcpy(h_info.data(), scratch.work().info, LUF_INFO_WORDS * sizeof(int), hipMemcpyDeviceToHost));
        h_info[LUF_INFO_WORDS - 1] = (int)(elapsed_ms * 1e4f);
        if (info) std::copy(h_info.begin(), h_info.end(), info);
        if (h_info[LUF_STATUS] == LUF_ERR_SINGULAR) throw std::runtime_error("singular basis");
        if (h_info[LUF_STATUS] == LUF_ERR_L_CAPACITY || h_info[LUF_STATUS] == LUF_ERR_U_CAPACITY || h_info[LUF_STATUS] == LUF_ERR_INVERSE_CAPACITY)
            throw std::invalid_argument("capacity too small");
        if (h_info[LUF_STATUS] != LUF_OK) throw std::runtime_error("device LU factorisation failed with status " + std::to_string(h_info[LUF_STATUS]));
        std::vector<int> hi(std::max<size_t>(cap, (size_t)m + 1));
        std::vector<double> hd(std::max<size_t>(cap, (size_t)m + 1));
        auto get_i = [&](const int* dev, size_t n, auto* dst) {
            if (!dst || n == 0) return;
            RELP_HIP(hipMemcpy(hi.data(), dev, n * sizeof(int), hipMemcpyDeviceToHost));
            for (size_t k = 0; k < n; ++k) dst[k] = hi[k];
        };
        auto get_d = [&](const double* dev, size_t n, double* dst) {
            if (!dst || n == 0) return;
            RELP_HIP(hipMemcpy(dst, dev, n * sizeof(double), hipMemcpyDeviceToHost));
        };
        const size_t nl = (size_t)h_info[inverted ? LUF_NNZ_LI : LUF_NNZ_L], nu = (size_t)h_info[inverted ? LUF_NNZ_UI : LUF_NNZ_U];
        get_i(b.rowpos, m, row_permutation);
        get_i(b.colpos, m, column_permutation);
        if (inverted) {  // L^-1 (strict) and U^-1 (with its diagonal) by rows; the diagonal array of this form is ones
            const LuInverseWork& iw = scratch.inverse_work();
            get_i(iw.csr_start[0], m + 1, lower_start);
            get_i(iw.csr_idx[0], nl, lower_column);
            get_d(iw.csr_val[0], nl, lower_value);
            get_i(iw.csr_start[1], m + 1, upper_start);
            get_i(iw.csr_idx[1], nu, upper_column);
            get_d(iw.csr_val[1], nu, upper_value);
            if (upper_diagonal) std::fill(upper_diagonal, upper_diagonal + m, 1.0);
        } else {
            get_i(b.ls, m + 1, lower_start);
            get_i(b.lc, nl, lower_column);
            get_d(b.lv, nl, lower_value);
            get_i(b.us, m + 1, upper_start);
            get_i(b.uc, nu, upper_column);
            get_d(b.uv, nu, upper_value);
            get_d(b.diag, m, upper_diagonal);
        }
    });
}

}  // extern "C"

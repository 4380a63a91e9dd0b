// extern "C" boundary (include/relp_amd.h).  Plain pointers and sizes only; exceptions are mapped to status codes.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <new>
#include <sstream>

#include "network.hpp"
#include "presolve.hpp"
#include "solver.hpp"

namespace relp {
void exact_finish_entries(int device, int limbs, int count, const unsigned long long* T, const int* carry, const int* words, int shift, int flip,
                          unsigned long long* N_out, int* bits_out);
void exact_words_test(int device, int limbs, int mode, int count, const unsigned long long* a, const unsigned long long* b, unsigned long long* out);
void grid_barrier_test(int device, int grid, int rounds, int reads, int mode, long long limit_ticks, long long* out8);
double exact_tile_bench(int device, int limbs, int tiles, int nb64, int terms, int shift);
}
using namespace relp;

struct relp_model {
    StandardForm form;
};

struct relp_handle {
    relp_options options;
    Solver* solver = nullptr;
    std::string error;
};

namespace {

template <class F>
int32_t guarded(relp_handle* h, F&& f) {
    try {
        const TuningScope tuning(h ? tuning_of(h->options) : thread_tuning());  // the handle's switches, for the helpers that have no handle in reach
        f();
        return RELP_OK;
    } catch (const DeviceError& e) {
        if (h) h->error = e.what();
        return RELP_ERR_DEVICE;
    } catch (const RatOverflow& e) {
        if (h) h->error = e.what();
        return RELP_ERR_OVERFLOW;
    } catch (const std::invalid_argument& e) {
        if (h) h->error = e.what();
        return RELP_ERR_ARGUMENT;
    } catch (const std::exception& e) {
        if (h) h->error = e.what();
        return RELP_ERR_STATE;
    }
}

Rat make_rat(int64_t n, int64_t d) { return Rat((i128)n, (i128)d); }

}  // namespace

extern "C" {

const char* relp_version(void) { return "relp_amd 0.1 gfx950"; }

// The library's defaults in the library's own struct (every field of every round).
static void library_defaults(relp_options* o) {
    std::memset(o, 0, sizeof(*o));
    o->device = 0;
    o->pivot_rule = RELP_PIVOT_STEEPEST_EDGE;  // two_phase/mod.rs:57,68,107
    o->polish_period = 256;
    o->pivots_per_launch = 64;
    o->max_pivots = 0;
    o->tol_dual = 1e-9;
    o->tol_pivot = 1e-9;
    o->harris_delta = 1e-9;
    o->tol_feasible = 1e-7;
    o->certify = 0;
    o->use_graph = 1;
    o->verbose = 0;
    o->implicit_bounds = 0;
    o->carry = RELP_CARRY_EXPLICIT;
    o->refactor_period = 0;
    o->lu_pivot_threshold = 0.0;
    o->ratio_rule = RELP_RATIO_AUTO;   // (round 6: the reference's ratio test where the data are small integers, else Harris)
    o->crash = 0;
    o->dense_storage = RELP_DENSE_NARROWEST;
    o->pivot_kernels = 0;
    o->product_form = 0;
    o->ftran_min_nnz = 0;
    o->lu_refactor = RELP_REFACTOR_AUTO;
    o->struct_size = (int32_t)sizeof(relp_options);  // (the switches and sizes appended in round 5 stay 0: the library's choices)
}

// Round 6 (advisor, medium): the CALLER states the size of the struct it compiled.  Only that many bytes are written -- a caller built
// against an older, shorter header gets no byte past the end of its struct -- and struct_size is the caller's size, which relp_create /
// relp_batch_create then read.  The sizes a header has ever had are the only ones accepted (a field is never copied in half).
int32_t relp_options_default_sized(relp_options* o, int32_t caller_size) {
    if (!o || !known_options_size(caller_size)) return RELP_ERR_ARGUMENT;
    relp_options full;
    library_defaults(&full);
    full.struct_size = caller_size;
    std::memcpy(o, &full, (size_t)caller_size);
    return RELP_OK;
}
// The exported symbol of rounds 1-5 for callers that bind by name (ctypes, a Rust `extern` block) and are built against THIS header: it
// assumes the library's size.  C callers never reach it: the header maps the name to relp_options_default_sized(o, sizeof(relp_options)).
#undef relp_options_default
int32_t relp_options_default(relp_options* o) { return relp_options_default_sized(o, (int32_t)sizeof(relp_options)); }


// ---- host-only model ------------------------------------------------------------------------------------
int32_t relp_model_from_mps_ex(const char* path, int32_t fixed_format, int32_t presolve, relp_model** out, char* error,
                               int32_t error_capacity) {
    if (!path || !out) return RELP_ERR_ARGUMENT;
    *out = nullptr;
    auto fail = [&](const std::string& what, int32_t code) {
        if (error && error_capacity > 0) {
            std::strncpy(error, what.c_str(), error_capacity - 1);
            error[error_capacity - 1] = 0;
        }
        return code;
    };
    std::ifstream in(path);
    if (!in) return fail(std::string("cannot open ") + path, RELP_ERR_PARSE);
    std::stringstream buffer;
    buffer << in.rdbuf();
    try {
        std::unique_ptr<relp_model> model(new relp_model());
        model->form = load_mps(buffer.str(), fixed_format != 0, presolve != 0);
        *out = model.release();
        return RELP_OK;
    } catch (const RatOverflow& e) {
        return fail(e.what(), RELP_ERR_OVERFLOW);
    } catch (const PresolveInfeasible& e) {
        return fail(e.what(), RELP_ERR_STATE);
    } catch (const PresolveUnbounded& e) {
        return fail(e.what(), RELP_ERR_STATE);
    } catch (const std::exception& e) {
        return fail(e.what(), RELP_ERR_PARSE);
    }
}
int32_t relp_model_from_general_form(int32_t maximize, int32_t nr_rows, int32_t nr_columns, const int64_t* column_start,
                                     const int32_t* row_index, const int64_t* value_num, const int64_t* value_den,
                                     const int32_t* row_kind, const int64_t* range_num, const int64_t* range_den,
                                     const int64_t* b_num, const int64_t* b_den, const int64_t* cost_num, const int64_t* cost_den,
                                     const uint8_t* has_lower, const int64_t* lower_num, const int64_t* lower_den,
                                     const uint8_t* has_upper, const int64_t* upper_num, const int64_t* upper_den,
                                     int64_t fixed_cost_num, int64_t fixed_cost_den, int32_t presolve, relp_model** out,
                                     char* error, int32_t error_capacity) {
    if (!out) return RELP_ERR_ARGUMENT;
    *out = nullptr;
    auto fail = [&](const std::string& what, int32_t code) {
        if (error && error_capacity > 0) {
            std::strncpy(error, what.c_str(), error_capacity - 1);
            error[error_capacity - 1] = 0;
        }
        return code;
    };
    if (nr_rows < 1 || nr_columns < 1 || !column_start || !row_kind || !b_num || !b_den || !cost_num || !cost_den || !has_lower ||
        !has_upper || fixed_cost_den == 0)
        return fail("general form: missing array or empty problem", RELP_ERR_ARGUMENT);
    try {
        auto rational = [](int64_t n, int64_t d) {
            if (d == 0) throw std::invalid_argument("general form: zero denominator");
            return Rat((i128)n, (i128)d);
        };
        GeneralInput general;
        general.maximize = maximize != 0;
        general.fixed_cost = rational(fixed_cost_num, fixed_cost_den);
        for (int32_t j = 0; j < nr_columns; ++j) {
            GeneralVariable v;
            v.cost = rational(cost_num[j], cost_den[j]);
            v.has_lower = has_lower[j] != 0;
            v.has_upper = has_upper[j] != 0;
            if (v.has_lower) v.lower = rational(lower_num[j], lower_den[j]);
            if (v.has_upper) v.upper = rational(upper_num[j], upper_den[j]);
            if (v.has_lower && v.has_upper && v.lower > v.upper) throw std::invalid_argument("general form: lower bound above upper bound");
            general.variables.push_back(v);
            SparseColumn column;
            if (column_start[j] > column_start[j + 1]) throw std::invalid_argument("general form: column_start must not decrease");
            for (int64_t e = column_start[j]; e < column_start[j + 1]; ++e) column.push(row_index[e], rational(value_num[e], value_den[e]));
            general.columns.push_back(column);
            general.column_names.push_back("X" + std::to_string(j));
        }
        for (int32_t i = 0; i < nr_rows; ++i) {
            if (row_kind[i] < 0 || row_kind[i] > 3) throw std::invalid_argument("general form: unknown row kind");
            general.kind.push_back((RowKind)row_kind[i]);
            Rat r(0);
            if (row_kind[i] == RANGE) {
                if (!range_num || !range_den) throw std::invalid_argument("general form: range rows without ranges");
                r = rational(range_num[i], range_den[i]);
                if (r.sign() < 0) throw std::invalid_argument("general form: negative range");
            }
            general.range.push_back(r);
            general.b.push_back(rational(b_num[i], b_den[i]));
        }
        std::unique_ptr<relp_model> model(new relp_model());
        model->form = standardize_general_form(std::move(general), presolve != 0);
        *out = model.release();
        return RELP_OK;
    } catch (const RatOverflow& e) {
        return fail(e.what(), RELP_ERR_OVERFLOW);
    } catch (const PresolveInfeasible& e) {
        return fail(e.what(), RELP_ERR_STATE);
    } catch (const PresolveUnbounded& e) {
        return fail(e.what(), RELP_ERR_STATE);
    } catch (const std::invalid_argument& e) {
        return fail(e.what(), RELP_ERR_ARGUMENT);
    } catch (const std::exception& e) {
        return fail(e.what(), RELP_ERR_PARSE);
    }
}
int32_t relp_model_from_provider(const relp_provider* provider, relp_model** out, char* error, int32_t error_capacity) {
    if (!out) return RELP_ERR_ARGUMENT;
    *out = nullptr;
    auto fail = [&](const std::string& what, int32_t code) {
        if (error && error_capacity > 0) {
            std::strncpy(error, what.c_str(), error_capacity - 1);
            error[error_capacity - 1] = 0;
        }
        return code;
    };
    if (!provider || !provider->column || !provider->cost_value || !provider->right_hand_side || provider->nr_rows < 1 ||
        provider->nr_columns < 1)
        return fail("provider: missing callback or empty problem", RELP_ERR_ARGUMENT);
    try {
        const int m = provider->nr_rows, n = provider->nr_columns;
        auto rational = [](int64_t num, int64_t den) {
            if (den == 0) throw std::invalid_argument("provider: zero denominator");
            return Rat((i128)num, (i128)den);
        };
        std::unique_ptr<relp_model> model(new relp_model());
        StandardForm& form = model->form;
        MatrixData& data = form.data;
        data.nr_equality = m;
        data.constraints.resize(n);
        data.variables.resize(n);
        std::vector<int32_t> rows(16);
        std::vector<int64_t> nums(16), dens(16);
        for (int j = 0; j < n; ++j) {
            int32_t nnz = provider->column(provider->user, j, (int32_t)rows.size(), rows.data(), nums.data(), dens.data());
            if (nnz > (int32_t)rows.size()) {
                rows.resize(nnz); nums.resize(nnz); dens.resize(nnz);
                nnz = provider->column(provider->user, j, nnz, rows.data(), nums.data(), dens.data());
            }
            if (nnz < 0 || nnz > (int32_t)rows.size()) throw std::invalid_argument("provider: bad column length");
            for (int32_t e = 0; e < nnz; ++e) {
                if (rows[e] < 0 || rows[e] >= m || (e > 0 && rows[e] <= rows[e - 1])) throw std::invalid_argument("provider: column rows must ascend within [0, nr_rows)");
                const Rat v = rational(nums[e], dens[e]);
                if (v.is_zero()) throw std::invalid_argument("provider: explicit zero in a column");
                data.constraints[j].push(rows[e], v);
            }
            int64_t cn = 0, cd = 1;
            provider->cost_value(provider->user, j, &cn, &cd);
            data.variables[j].cost = rational(cn, cd);
            form.column_names.push_back("X" + std::to_string(j));
            form.active_to_original.push_back(j);
        }
        form.all_column_names = form.column_names;
        form.nr_original = n;
        form.free_negative_part.assign(n, -1);
        std::vector<int64_t> bn(m, 0), bd(m, 1);
        provider->right_hand_side(provider->user, bn.data(), bd.data());
        for (int i = 0; i < m; ++i) {
            data.b.push_back(rational(bn[i], bd[i]));
            if (data.b.back().sign() < 0) throw std::invalid_argument("provider: negative right-hand side");
        }
        data.finalize();
        if (provider->pivot_element_indices) {
            std::vector<int32_t> pr(m), pc(m);
            const int32_t count = provider->pivot_element_indices(provider->user, m, pr.data(), pc.data());
            if (count < 0 || count > m) throw std::invalid_argument("provider: bad number of initial pivots");
            std::vector<char> row_used(m, 0), column_used(n, 0);
            for (int32_t k = 0; k < count; ++k) {
                const int r = pr[k], c = pc[k];
                if (r < 0 || r >= m || c < 0 || c >= n || row_used[r] || column_used[c]) throw std::invalid_argument("provider: bad initial pivot");
                const SparseColumn& column = data.constraints[c];
                if (column.nnz() != 1 || column.index[0] != r || !(column.value[0] == Rat(1)))
                    throw std::invalid_argument("provider: an initial pivot column must be the unit vector of its row");
                row_used[r] = column_used[c] = 1;
                data.provider_pivots.push_back({r, c});
            }
        }
        *out = model.release();
        return RELP_OK;
    } catch (const RatOverflow& e) {
        return fail(e.what(), RELP_ERR_OVERFLOW);
    } catch (const std::exception& e) {
        return fail(e.what(), RELP_ERR_ARGUMENT);
    }
}
int32_t relp_model_from_mps(const char* path, int32_t fixed_format, relp_model** out, char* error, int32_t error_capacity) {
    return relp_model_from_mps_ex(path, fixed_format, 0, out, error, error_capacity);
}
int32_t relp_model_original_variables(const relp_model* model, int32_t* nr_original, int32_t* nr_removed) {
    if (!model) return RELP_ERR_ARGUMENT;
    if (nr_original) *nr_original = (int32_t)model->form.nr_file_variables();
    if (nr_removed) *nr_removed = (int32_t)model->form.removed.size();
    return RELP_OK;
}
static int32_t model_from_graph(bool max_flow, int32_t nr_vertices, int32_t nr_arcs, const int32_t* tail, const int32_t* head,
                                const int64_t* num, const int64_t* den, int32_t s, int32_t t, relp_model** out, char* error,
                                int32_t error_capacity) {
    if (!out || nr_arcs < 0 || (nr_arcs > 0 && (!tail || !head || !num || !den))) return RELP_ERR_ARGUMENT;
    *out = nullptr;
    auto fail = [&](const std::string& what, int32_t code) {
        if (error && error_capacity > 0) {
            std::strncpy(error, what.c_str(), error_capacity - 1);
            error[error_capacity - 1] = 0;
        }
        return code;
    };
    try {
        std::vector<Arc> arcs((size_t)nr_arcs);
        for (int32_t k = 0; k < nr_arcs; ++k) {
            if (den[k] == 0) return fail("zero denominator", RELP_ERR_ARGUMENT);
            arcs[k] = Arc{tail[k], head[k], make_rat(num[k], den[k])};
        }
        std::unique_ptr<relp_model> model(new relp_model());
        model->form = max_flow ? make_max_flow(nr_vertices, arcs, s, t) : make_shortest_path(nr_vertices, arcs, s, t);
        *out = model.release();
        return RELP_OK;
    } catch (const RatOverflow& e) {
        return fail(e.what(), RELP_ERR_OVERFLOW);
    } catch (const std::exception& e) {
        return fail(e.what(), RELP_ERR_ARGUMENT);
    }
}
int32_t relp_model_max_flow(int32_t nr_vertices, int32_t nr_arcs, const int32_t* tail, const int32_t* head,
                            const int64_t* capacity_num, const int64_t* capacity_den, int32_t s, int32_t t,
                            relp_model** out, char* error, int32_t error_capacity) {
    return model_from_graph(true, nr_vertices, nr_arcs, tail, head, capacity_num, capacity_den, s, t, out, error, error_capacity);
}
int32_t relp_model_shortest_path(int32_t nr_vertices, int32_t nr_arcs, const int32_t* tail, const int32_t* head,
                                 const int64_t* length_num, const int64_t* length_den, int32_t s, int32_t t,
                                 relp_model** out, char* error, int32_t error_capacity) {
    return model_from_graph(false, nr_vertices, nr_arcs, tail, head, length_num, length_den, s, t, out, error, error_capacity);
}
int32_t relp_model_free(relp_model* model) {
    delete model;
    return RELP_OK;
}
static int32_t model_dimensions(const MatrixData& md, int32_t* nr_rows, int32_t* nr_columns, int32_t* nr_constraints,
                                int32_t* nr_structural, int64_t* nnz, int32_t group_counts[4]) {
    if (nr_rows) *nr_rows = md.nr_rows();
    if (nr_columns) *nr_columns = md.nr_columns();
    if (nr_constraints) *nr_constraints = md.nr_constraints();
    if (nr_structural) *nr_structural = md.nr_normal_variables();
    if (nnz) {
        int64_t total = 0;
        for (const auto& c : md.constraints) total += (int64_t)c.nnz();
        *nnz = total;
    }
    if (group_counts) {
        group_counts[0] = md.nr_equality;
        group_counts[1] = md.nr_range;
        group_counts[2] = md.nr_upper;
        group_counts[3] = md.nr_lower;
    }
    return RELP_OK;
}
int32_t relp_model_dimensions(const relp_model* model, int32_t* nr_rows, int32_t* nr_columns, int32_t* nr_constraints,
                              int32_t* nr_structural, int64_t* nnz, int32_t group_counts[4]) {
    if (!model) return RELP_ERR_ARGUMENT;
    return model_dimensions(model->form.data, nr_rows, nr_columns, nr_constraints, nr_structural, nnz, group_counts);
}
int32_t relp_model_column(const relp_model* model, int32_t j, int32_t capacity, int32_t* count, int32_t* row_index, double* value) {
    if (!model || !count || j < 0 || j >= model->form.data.nr_columns()) return RELP_ERR_ARGUMENT;
    SparseColumn c = model->form.data.column(j);
    *count = (int32_t)c.nnz();
    for (int32_t e = 0; e < *count && e < capacity; ++e) {
        if (row_index) row_index[e] = c.index[e];
        if (value) value[e] = c.value[e].to_double();
    }
    return RELP_OK;
}
int32_t relp_model_column_exact(const relp_model* model, int32_t j, int32_t capacity, int32_t* count, int32_t* row_index,
                                int64_t* num, int64_t* den) {
    if (!model || !count || j < 0 || j >= model->form.data.nr_columns()) return RELP_ERR_ARGUMENT;
    SparseColumn c = model->form.data.column(j);
    *count = (int32_t)c.nnz();
    for (int32_t e = 0; e < *count && e < capacity; ++e) {
        const Rat& v = c.value[e];
        if (v.n > INT64_MAX || v.n < INT64_MIN || v.d > INT64_MAX) return RELP_ERR_OVERFLOW;
        if (row_index) row_index[e] = c.index[e];
        if (num) num[e] = (int64_t)v.n;
        if (den) den[e] = (int64_t)v.d;
    }
    return RELP_OK;
}
int32_t relp_model_cost(const relp_model* model, int32_t j, double* cost) {
    if (!model || !cost || j < 0 || j >= model->form.data.nr_columns()) return RELP_ERR_ARGUMENT;
    *cost = model->form.data.cost_value(j).to_double();
    return RELP_OK;
}
int32_t relp_model_right_hand_side(const relp_model* model, double* rhs) {
    if (!model || !rhs) return RELP_ERR_ARGUMENT;
    auto values = model->form.data.right_hand_side();
    for (size_t i = 0; i < values.size(); ++i) rhs[i] = values[i].to_double();
    return RELP_OK;
}
int32_t relp_model_initial_pivots(const relp_model* model, int32_t capacity, int32_t* count, int32_t* rows, int32_t* columns) {
    if (!model || !count) return RELP_ERR_ARGUMENT;
    auto pivots = model->form.data.pivot_element_indices();
    *count = (int32_t)pivots.size();
    for (int32_t k = 0; k < *count && k < capacity; ++k) {
        if (rows) rows[k] = pivots[k].first;
        if (columns) columns[k] = pivots[k].second;
    }
    return RELP_OK;
}
int32_t relp_model_fixed_cost(const relp_model* model, double* fixed_cost) {
    if (!model || !fixed_cost) return RELP_ERR_ARGUMENT;
    *fixed_cost = model->form.fixed_cost.to_double();
    return RELP_OK;
}

int32_t relp_load_model(relp_handle* h, const relp_model* model) {
    if (!h || !h->solver || !model) return RELP_ERR_ARGUMENT;
    return guarded(h, [&] {
        StandardForm copy = model->form;
        h->solver->load(std::move(copy));
    });
}

int32_t relp_create(const relp_options* options, relp_handle** out) {
    if (!out) return RELP_ERR_ARGUMENT;
    *out = nullptr;
    relp_handle* h = new (std::nothrow) relp_handle();
    if (!h) return RELP_ERR_STATE;
    if (adopt_options(options, &h->options) != RELP_OK) {
        delete h;
        return RELP_ERR_ARGUMENT;
    }
    int32_t status = guarded(h, [&] { h->solver = new Solver(h->options); });
    if (status != RELP_OK) {
        // keep the message reachable: the handle is returned only on success
        delete h;
        return status;
    }
    *out = h;
    return RELP_OK;
}

int32_t relp_destroy(relp_handle* h) {
    if (!h) return RELP_ERR_ARGUMENT;
    delete h->solver;
    delete h;
    return RELP_OK;
}

const char* relp_last_error(const relp_handle* h) { return h ? h->error.c_str() : "null handle"; }

int32_t relp_load_matrix_data(relp_handle* h, int32_t nr_constraints, int32_t nr_variables,
                              const int64_t* column_start, const int32_t* row_index,
                              const int64_t* value_num, const int64_t* value_den,
                              const int64_t* b_num, const int64_t* b_den,
                              const int64_t* cost_num, const int64_t* cost_den,
                              const uint8_t* has_upper, const int64_t* upper_num, const int64_t* upper_den,
                              const int64_t* range_num, const int64_t* range_den,
                              int32_t nr_equality, int32_t nr_range, int32_t nr_upper, int32_t nr_lower,
                              int64_t fixed_cost_num, int64_t fixed_cost_den) {
    if (!h || !column_start || nr_constraints < 0 || nr_variables < 0) return RELP_ERR_ARGUMENT;
    if (nr_equality + nr_range + nr_upper + nr_lower != nr_constraints) {
        h->error = "row group counts do not add up to nr_constraints";
        return RELP_ERR_ARGUMENT;
    }
    return guarded(h, [&] {
        StandardForm form;
        MatrixData& md = form.data;
        md.nr_equality = nr_equality;
        md.nr_range = nr_range;
        md.nr_upper = nr_upper;
        md.nr_lower = nr_lower;
        md.constraints.resize(nr_variables);
        md.variables.resize(nr_variables);
        for (int j = 0; j < nr_variables; ++j) {
            int previous = -1;
            for (int64_t e = column_start[j]; e < column_start[j + 1]; ++e) {
                if (row_index[e] <= previous || row_index[e] >= nr_constraints) throw std::invalid_argument("rows of a column must be sorted, unique and in range");
                previous = row_index[e];
                Rat v = make_rat(value_num[e], value_den ? value_den[e] : 1);
                if (v.is_zero()) throw std::invalid_argument("explicit zero in sparse column");
                md.constraints[j].push(row_index[e], v);
            }
            md.variables[j].cost = make_rat(cost_num[j], cost_den ? cost_den[j] : 1);
            if (has_upper && has_upper[j]) {
                md.variables[j].has_upper = true;
                md.variables[j].upper = make_rat(upper_num[j], upper_den ? upper_den[j] : 1);
            }
        }
        md.b.resize(nr_constraints);
        for (int i = 0; i < nr_constraints; ++i) {
            md.b[i] = make_rat(b_num[i], b_den ? b_den[i] : 1);
            if (md.b[i].sign() < 0) throw std::invalid_argument("b must be non-negative (general_form/mod.rs:592-618)");
        }
        for (int i = 0; i < nr_range; ++i) md.ranges.push_back(make_rat(range_num[i], range_den ? range_den[i] : 1));
        md.finalize();
        form.fixed_cost = make_rat(fixed_cost_num, fixed_cost_den ? fixed_cost_den : 1);
        form.nr_original = nr_variables;
        form.free_negative_part.assign(nr_variables, -1);
        h->solver->load(std::move(form));
    });
}

int32_t relp_load_dense_le(relp_handle* h, int32_t m, int32_t n, const int64_t* a, const int64_t* b, const int64_t* cost) {
    if (!h || !h->solver || m < 1 || n < 1 || !a || !b || !cost) return RELP_ERR_ARGUMENT;
    return guarded(h, [&] {
        StandardForm form;
        MatrixData& md = form.data;
        md.nr_upper = m;
        md.constraints.resize(n);
        md.variables.resize(n);
        for (int j = 0; j < n; ++j) {
            SparseColumn& c = md.constraints[j];
            c.index.reserve(m);
            c.value.reserve(m);
            for (int i = 0; i < m; ++i) {
                const int64_t v = a[(size_t)j * m + i];
                if (v != 0) c.push(i, Rat((long long)v));
            }
            md.variables[j].cost = Rat((long long)cost[j]);
        }
        md.b.resize(m);
        for (int i = 0; i < m; ++i) {
            if (b[i] < 0) throw std::invalid_argument("b must be non-negative");
            md.b[i] = Rat((long long)b[i]);
        }
        md.finalize();
        form.nr_original = n;
        form.free_negative_part.assign(n, -1);
        h->solver->load(std::move(form));
    });
}

int32_t relp_load_mps_ex(relp_handle* h, const char* path, int32_t fixed_format, int32_t presolve) {
    if (!h || !path) return RELP_ERR_ARGUMENT;
    std::ifstream in(path);
    if (!in) {
        h->error = std::string("cannot open ") + path;
        return RELP_ERR_PARSE;
    }
    std::stringstream buffer;
    buffer << in.rdbuf();
    StandardForm form;
    try {
        form = load_mps(buffer.str(), fixed_format != 0, presolve != 0);
    } catch (const RatOverflow& e) {
        h->error = e.what();
        return RELP_ERR_OVERFLOW;
    } catch (const PresolveInfeasible& e) {
        h->error = e.what();
        return RELP_ERR_STATE;
    } catch (const PresolveUnbounded& e) {
        h->error = e.what();
        return RELP_ERR_STATE;
    } catch (const std::exception& e) {
        h->error = e.what();
        return RELP_ERR_PARSE;
    }
    return guarded(h, [&] { h->solver->load(std::move(form)); });
}
int32_t relp_load_mps(relp_handle* h, const char* path, int32_t fixed_format) { return relp_load_mps_ex(h, path, fixed_format, 0); }

#define REQUIRE_LOADED(h)                                   \
    if (!(h) || !(h)->solver) return RELP_ERR_ARGUMENT;     \
    if (!(h)->solver->loaded()) {                           \
        const_cast<relp_handle*>(h)->error = "no LP loaded"; \
        return RELP_ERR_STATE;                              \
    }

int32_t relp_get_dimensions(const relp_handle* h, int32_t* nr_rows, int32_t* nr_columns, int32_t* nr_constraints,
                            int32_t* nr_structural, int32_t* nr_artificial, int64_t* nnz) {
    REQUIRE_LOADED(h);
    if (nr_artificial) *nr_artificial = h->solver->n_art();
    return model_dimensions(h->solver->form().data, nr_rows, nr_columns, nr_constraints, nr_structural, nnz, nullptr);
}

int32_t relp_get_column(const relp_handle* h, int32_t j, int32_t capacity, int32_t* count, int32_t* row_index, double* value) {
    REQUIRE_LOADED(h);
    const MatrixData& md = h->solver->form().data;
    if (j < 0 || j >= md.nr_columns() || !count) return RELP_ERR_ARGUMENT;
    SparseColumn c = md.column(j);
    *count = (int32_t)c.nnz();
    for (int32_t e = 0; e < *count && e < capacity; ++e) {
        if (row_index) row_index[e] = c.index[e];
        if (value) value[e] = c.value[e].to_double();
    }
    return RELP_OK;
}

int32_t relp_get_cost(const relp_handle* h, int32_t j, double* cost) {
    REQUIRE_LOADED(h);
    const MatrixData& md = h->solver->form().data;
    if (j < 0 || j >= md.nr_columns() || !cost) return RELP_ERR_ARGUMENT;
    *cost = md.cost_value(j).to_double();
    return RELP_OK;
}

int32_t relp_get_right_hand_side(const relp_handle* h, double* rhs) {
    REQUIRE_LOADED(h);
    if (!rhs) return RELP_ERR_ARGUMENT;
    auto values = h->solver->form().data.right_hand_side();
    for (size_t i = 0; i < values.size(); ++i) rhs[i] = values[i].to_double();
    return RELP_OK;
}

int32_t relp_get_initial_pivots(const relp_handle* h, int32_t capacity, int32_t* count, int32_t* rows, int32_t* columns) {
    REQUIRE_LOADED(h);
    if (!count) return RELP_ERR_ARGUMENT;
    auto pivots = h->solver->form().data.pivot_element_indices();
    *count = (int32_t)pivots.size();
    for (int32_t k = 0; k < *count && k < capacity; ++k) {
        if (rows) rows[k] = pivots[k].first;
        if (columns) columns[k] = pivots[k].second;
    }
    return RELP_OK;
}

int32_t relp_solve_relaxation(relp_handle* h, relp_result* result) {
    REQUIRE_LOADED(h);
    return guarded(h, [&] {
        h->solver->last_error.clear();
        h->solver->solve(result);
        if (!h->solver->last_error.empty()) h->error = h->solver->last_error;  // e.g. why an exact certificate was not obtained
    });
}

int32_t relp_get_solution(const relp_handle* h, double* x) {
    REQUIRE_LOADED(h);
    if (!x) return RELP_ERR_ARGUMENT;
    h->solver->get_solution(x);
    return RELP_OK;
}

int32_t relp_get_original_solution(const relp_handle* h, int32_t capacity, double* x, int32_t* count) {
    REQUIRE_LOADED(h);
    const StandardForm& form = h->solver->form();
    const int32_t total = (int32_t)form.nr_file_variables();
    if (count) *count = total;
    if (!x || capacity < total) return total == 0 ? RELP_OK : RELP_ERR_ARGUMENT;
    std::vector<double> standardised((size_t)form.data.nr_normal_variables());
    h->solver->get_solution(standardised.data());
    const std::vector<double> original = form.original_solution(standardised);
    for (int32_t j = 0; j < total; ++j) x[j] = original[j];
    return RELP_OK;
}

// Exact back-mapping of general_form/mod.rs:753-771, 840-934 (the f64 twin is StandardForm::original_solution).
static std::vector<BigRat> original_solution_exact(const StandardForm& form, const std::vector<BigRat>& standardised) {
    const bool identity = form.active_to_original.empty();
    std::vector<BigRat> out((size_t)form.nr_file_variables());
    std::vector<char> known(out.size(), 0);
    for (int j = 0; j < form.nr_original; ++j) {
        BigRat x = standardised[j];
        if (j < (int)form.free_negative_part.size() && form.free_negative_part[j] >= 0) x -= standardised[form.free_negative_part[j]];
        x -= BigRat(form.data.variables[j].shift);
        if (form.data.variables[j].flipped) x = -x;
        const int original = identity ? j : form.active_to_original[j];
        out[original] = x;
        known[original] = 1;
    }
    bool progress = true;
    while (progress) {
        progress = false;
        for (const auto& [original, how] : form.removed) {
            if (known[original]) continue;
            bool ready = true;
            BigRat value(how.constant);
            if (how.function_of_others)
                for (const auto& [k, c] : how.coefficients) {
                    if (!known[k]) { ready = false; break; }
                    value -= BigRat(c) * out[k];
                }
            if (ready) {
                out[original] = value;
                known[original] = 1;
                progress = true;
            }
        }
    }
    return out;
}

int32_t relp_get_solution_exact(const relp_handle* h, int32_t original, int32_t capacity, int32_t* count, int32_t* index,
                                char* buffer, int64_t buffer_capacity, int64_t* length) {
    REQUIRE_LOADED(h);
    if (!count || capacity < 0 || buffer_capacity < 0) return RELP_ERR_ARGUMENT;
    const Solver& sv = *h->solver;
    if (!sv.exact_primal || sv.last_result.kind != RELP_RESULT_FINITE_OPTIMUM || !sv.last_result.certified) {
        const_cast<relp_handle*>(h)->error = "no exact solution (set options.certify and solve to a certified finite optimum)";
        return RELP_ERR_STATE;
    }
    try {
        const StandardForm& form = sv.form();
        const int n_structural = form.data.nr_normal_variables();
        std::vector<std::pair<int, std::string>> values;
        const auto basics = exact_primal_values(*sv.exact_primal);  // every provider column, slacks included
        if (!original) {
            for (const auto& entry : basics)  // reconstruct_solution (matrix_data.rs:402-411): the slack columns are dropped
                if (entry.first < n_structural) values.push_back(entry);
        } else {
            std::vector<BigRat> standardised((size_t)n_structural);
            for (const auto& [j, text] : basics)
                if (j < n_structural) standardised[j] = BigRat::parse(text);
            const std::vector<BigRat> full = original_solution_exact(form, standardised);
            for (size_t j = 0; j < full.size(); ++j)
                if (!full[j].is_zero()) values.push_back({(int)j, full[j].to_string()});
        }
        *count = (int32_t)values.size();
        int64_t needed = 0;
        for (const auto& entry : values) needed += (int64_t)entry.second.size() + 1;
        if (length) *length = needed;
        if (!index && !buffer) return RELP_OK;  // size query
        if (capacity < *count || buffer_capacity < needed || !index || !buffer) return RELP_ERR_ARGUMENT;
        int64_t at = 0;
        for (size_t k = 0; k < values.size(); ++k) {
            index[k] = values[k].first;
            std::memcpy(buffer + at, values[k].second.data(), values[k].second.size());
            at += (int64_t)values[k].second.size();
            buffer[at++] = k + 1 < values.size() ? '\n' : '\0';
        }
        return RELP_OK;
    } catch (const std::exception& e) {
        const_cast<relp_handle*>(h)->error = e.what();
        return RELP_ERR_NUMERICAL;
    }
}

int32_t relp_get_variable_name(const relp_handle* h, int32_t j, char* buffer, int32_t capacity, int32_t* length) {
    REQUIRE_LOADED(h);
    const StandardForm& form = h->solver->form();
    if (j < 0 || j >= form.nr_file_variables()) return RELP_ERR_ARGUMENT;
    const std::string name = !form.all_column_names.empty() ? form.all_column_names[j]
                             : j < (int)form.column_names.size() ? form.column_names[j] : "X" + std::to_string(j);
    if (length) *length = (int32_t)name.size();
    if (buffer && capacity > 0) {
        const int32_t nbytes = std::min<int32_t>((int32_t)name.size(), capacity - 1);
        std::memcpy(buffer, name.data(), nbytes);
        buffer[nbytes] = 0;
    }
    return RELP_OK;
}

int32_t relp_get_objective_exact(const relp_handle* h, char* buffer, int32_t capacity, int32_t* length) {
    REQUIRE_LOADED(h);
    const std::string& s = h->solver->exact_objective;
    if (length) *length = (int32_t)s.size();
    if (s.empty()) {
        const_cast<relp_handle*>(h)->error = "no exact objective (set options.certify and solve)";
        return RELP_ERR_STATE;
    }
    if (buffer && capacity > 0) {
        int32_t nbytes = std::min<int32_t>((int32_t)s.size(), capacity - 1);
        std::memcpy(buffer, s.data(), nbytes);
        buffer[nbytes] = 0;
    }
    return RELP_OK;
}

// (the name comes from the NAME record of an MPS file: untrusted text)
static std::string json_escaped(const std::string& text) {
    std::string out;
    for (unsigned char ch : text) {
        if (ch == '"' || ch == '\\') { out.push_back('\\'); out.push_back((char)ch); }
        else if (ch < 0x20) { char buf[8]; std::snprintf(buf, sizeof buf, "\\u%04x", ch); out += buf; }
        else out.push_back((char)ch);
    }
    return out;
}

// One JSON object per solved LP (metrics / observability -- the reference has none): dimensions, pivots per
// phase, refreshes of the inverse, wall times, pivots/s, algorithmic bytes moved by the loop, f64 and exact objective.
int32_t relp_get_record_json(const relp_handle* h, char* buffer, int32_t capacity, int32_t* length) {
    REQUIRE_LOADED(h);
    const Solver& sv = *h->solver;
    const relp_result& r = sv.last_result;
    const DeviceLP& d = sv.device();
    const relp_stats& st = sv.stats();
    const MatrixData& md = sv.form().data;
    long long nnz = 0;
    for (int j = 0; j < md.nr_columns(); ++j) nnz += (long long)md.column(j).nnz();
    const long long pivots = r.pivots_phase_one + r.pivots_phase_two;
    static const char* kinds[] = {"none", "finite_optimum", "infeasible", "unbounded", "iteration_limit"};
    std::ostringstream out;
    out.precision(17);
    static const char* presolve_states[] = {"off", "applied", "applied without the implied bounds beyond 126 bits", "dropped (did not fit the host model)"};
    out << "{\"name\": \"" << json_escaped(sv.form().name) << "\", \"presolve\": \"" << presolve_states[sv.form().presolve_state & 3] << "\", \"m\": " << md.nr_rows() << ", \"n\": " << md.nr_columns() << ", \"nnz\": " << nnz
        << ", \"device_rows\": " << d.m << ", \"artificials\": " << d.n_art << ", \"result\": \"" << kinds[r.kind >= 0 && r.kind <= 4 ? r.kind : 0]
        << "\", \"carry\": \"" << (h->options.carry == RELP_CARRY_LU ? "lu" : h->options.carry == RELP_CARRY_LU_INVERSE ? "lu_inverse" : "explicit") << "\", \"pivots_phase_one\": " << r.pivots_phase_one
        << ", \"pivots_phase_two\": " << r.pivots_phase_two << ", \"polishes\": " << r.polishes << ", \"refactors\": " << r.refactors
        << ", \"refactor_seconds\": " << r.refactor_seconds << ", \"ratio_rule\": \"" << (sv.ratio_textbook() ? "textbook" : "harris") << "\"" << ", \"lu_refactor\": \"" << (sv.refactors_asynchronously() ? "device, beside the pivots" : sv.refactors_on_device() ? "device" : "host")
        << "\", \"device_refactor_fallbacks\": " << sv.device_refactor_fallbacks()
        << ", \"async_refactors\": " << sv.async_refactors() << ", \"async_refactors_abandoned\": " << sv.async_refactors_abandoned() << ", \"async_worst_residual\": " << sv.async_worst_residual()
        << ", \"solve_seconds\": " << r.solve_seconds << ", \"certify_seconds\": " << r.certify_seconds
        << ", \"pivots_per_second\": " << (r.solve_seconds > 0 ? (double)pivots / r.solve_seconds : 0.0)
        << ", \"pricing_bytes_per_pivot\": " << st.price_bytes << ", \"inverse_bytes_per_pivot_bound\": " << (h->options.carry != RELP_CARRY_EXPLICIT ? 0 : st.update_bytes)
        << ", \"kernel_launches\": " << st.launches << ", \"certified\": " << (r.certified ? "true" : "false")
        << ", \"exact_repair_pivots\": " << r.exact_repair_pivots << ", \"objective\": ";
    if (r.kind == RELP_RESULT_FINITE_OPTIMUM) out << r.objective;
    else out << "null";
    out << ", \"objective_exact\": ";
    if (r.certified && !sv.exact_objective.empty()) out << "\"" << sv.exact_objective << "\"";
    else out << "null";
    out << "}";
    const std::string text = out.str();
    if (length) *length = (int32_t)text.size();
    if (buffer && capacity > 0) {
        const int32_t nbytes = std::min<int32_t>((int32_t)text.size(), capacity - 1);
        std::memcpy(buffer, text.data(), nbytes);
        buffer[nbytes] = 0;
    }
    return RELP_OK;
}

int32_t relp_solve_exact(relp_handle* h, int32_t first_limbs, int32_t max_limbs, int64_t max_pivots, relp_exact_result* result,
                         int32_t trace_capacity, int32_t* trace, char* objective, int32_t objective_capacity, int32_t* basis) {
    REQUIRE_LOADED(h);
    if (!result || first_limbs < 1 || max_limbs < first_limbs || max_limbs > 128 || trace_capacity < 0) return RELP_ERR_ARGUMENT;
    return guarded(h, [&] {
        std::memset(result, 0, sizeof(*result));
        std::vector<int> tr, final_basis;
        std::string text;
        std::vector<std::pair<int, long long>> survived;
        int status = 0, limbs = 0;
        long long p1 = 0, p2 = 0;
        int redundant = 0;
        h->solver->solve_exact(first_limbs, max_limbs, max_pivots, std::max(trace_capacity, 1), &status, &limbs, &p1, &p2, &tr, &text, &final_basis, &survived,
                               &redundant);
        result->status = status;
        result->redundant_rows = redundant;
        result->limbs = limbs;
        result->pivots_phase_one = p1;
        result->pivots_phase_two = p2;
        result->trace_entries = (int32_t)std::min<size_t>(tr.size() / 4, (size_t)trace_capacity);  // never more than were copied
        // (six slots: the LAST six widths tried -- 1, 2, ... 128 are eight, a run that needs the widest ones starts above one limb)
        const size_t first_kept = survived.size() > 6 ? survived.size() - 6 : 0;
        for (size_t k = first_kept; k < survived.size(); ++k) {
            result->limbs_tried[k - first_kept] = survived[k].first;
            result->pivots_survived[k - first_kept] = survived[k].second;
        }
        if (trace) std::memcpy(trace, tr.data(), std::min<size_t>(tr.size(), (size_t)4 * trace_capacity) * sizeof(int));
        if (objective && objective_capacity > 0) {
            const size_t nbytes = std::min<size_t>(text.size(), (size_t)objective_capacity - 1);
            std::memcpy(objective, text.data(), nbytes);
            objective[nbytes] = 0;
        }
        result->objective_length = (int32_t)text.size();
        if (basis && !final_basis.empty()) std::memcpy(basis, final_basis.data(), final_basis.size() * sizeof(int));
    });
}

int32_t relp_get_exact_counters(const relp_handle* h, relp_exact_width_record* records, int32_t capacity, int32_t* count) {
    REQUIRE_LOADED(h);
    if (!count || capacity < 0 || (capacity > 0 && !records)) return RELP_ERR_ARGUMENT;
    return guarded(const_cast<relp_handle*>(h), [&] {
        const std::vector<relp::ExactWidthRecord>& all = h->solver->exact_records();
        *count = (int32_t)all.size();
        for (size_t k = 0; k < all.size() && k < (size_t)capacity; ++k) {
            relp_exact_width_record& out = records[k];
            out.limbs = all[k].limbs;
            out.grid = all[k].grid;
            out.pivots_total_at_end = all[k].pivots_total_at_end;
            out.seconds = all[k].seconds;
            for (int t = 0; t < 10; ++t) out.step_seconds[t] = all[k].step_seconds[t];
            out.update_word_products_needed = all[k].update_products_needed;
            out.update_word_products_issued = all[k].update_products_issued;
        }
    });
}

int32_t relp_get_basis(const relp_handle* h, int32_t* basis) {
    REQUIRE_LOADED(h);
    if (!basis) return RELP_ERR_ARGUMENT;
    return guarded(const_cast<relp_handle*>(h), [&] { h->solver->get_basis(basis); });
}

int32_t relp_set_basis(relp_handle* h, const int32_t* basis_columns) {
    REQUIRE_LOADED(h);
    if (!basis_columns) return RELP_ERR_ARGUMENT;
    return guarded(h, [&] { h->solver->set_basis(basis_columns); });
}

int32_t relp_begin_phase_one(relp_handle* h) {
    REQUIRE_LOADED(h);
    return guarded(h, [&] { h->solver->begin_phase_one(); });
}
int32_t relp_begin_phase_two(relp_handle* h) {
    REQUIRE_LOADED(h);
    return guarded(h, [&] { h->solver->begin_phase_two(); });
}
int32_t relp_bi_ftran(relp_handle* h, int32_t nnz, const int32_t* rows, const double* values, double* out) {
    REQUIRE_LOADED(h);
    if (nnz < 0 || !out) return RELP_ERR_ARGUMENT;
    return guarded(h, [&] { h->solver->ftran(nnz, rows, values, out); });
}
int32_t relp_bi_btran(relp_handle* h, int32_t nnz, const int32_t* rows, const double* values, double* out) {
    REQUIRE_LOADED(h);
    if (nnz < 0 || !out) return RELP_ERR_ARGUMENT;
    return guarded(h, [&] { h->solver->btran(nnz, rows, values, out); });
}
int32_t relp_bi_row(relp_handle* h, int32_t row, double* out) {
    REQUIRE_LOADED(h);
    if (!out) return RELP_ERR_ARGUMENT;
    return guarded(h, [&] { h->solver->inverse_row(row, out); });
}
int32_t relp_price(relp_handle* h, int32_t* column, double* relative_cost) {
    REQUIRE_LOADED(h);
    if (!column || !relative_cost) return RELP_ERR_ARGUMENT;
    return guarded(h, [&] {
        int c;
        double v;
        h->solver->price(&c, &v);
        *column = c;
        *relative_cost = v;
    });
}
int32_t relp_relative_costs(relp_handle* h, double* out) {
    REQUIRE_LOADED(h);
    if (!out) return RELP_ERR_ARGUMENT;
    return guarded(h, [&] { h->solver->relative_costs(out); });
}
int32_t relp_get_gamma(relp_handle* h, double* out) {
    REQUIRE_LOADED(h);
    if (!out) return RELP_ERR_ARGUMENT;
    return guarded(h, [&] { h->solver->get_gamma(out); });
}
int32_t relp_ratio(relp_handle* h, int32_t column, int32_t* row, double* out_alpha) {
    REQUIRE_LOADED(h);
    if (!row) return RELP_ERR_ARGUMENT;
    return guarded(h, [&] {
        int r;
        h->solver->ratio(column, &r, out_alpha);
        *row = r;
    });
}
int32_t relp_bring_into_basis(relp_handle* h, int32_t column, int32_t row) {
    REQUIRE_LOADED(h);
    return guarded(h, [&] { h->solver->bring_into_basis(column, row); });
}
int32_t relp_get_last_pivot(relp_handle* h, int32_t* phase, int32_t* column, int32_t* row, int32_t* leaving) {
    REQUIRE_LOADED(h);
    return guarded(h, [&] {
        int ph = 0, q = -1, p = -1, l = -1;
        h->solver->last_pivot(&ph, &q, &p, &l);
        if (phase) *phase = ph;
        if (column) *column = q;
        if (row) *row = p;
        if (leaving) *leaving = l;
    });
}
int32_t relp_se_after_basis_update(relp_handle* h) {
    REQUIRE_LOADED(h);
    return guarded(h, [&] { h->solver->after_basis_update(); });
}
int32_t relp_refactor(relp_handle* h, double* residual_before) {
    REQUIRE_LOADED(h);
    return guarded(h, [&] {
        const double r = h->solver->refactor();
        if (residual_before) *residual_before = r;
    });
}
int32_t relp_iterate(relp_handle* h, int64_t count, int64_t* done, int32_t* stop_reason) {
    REQUIRE_LOADED(h);
    return guarded(h, [&] {
        int reason = 0;
        long long d = h->solver->iterate(count, &reason);
        if (done) *done = d;
        if (stop_reason) *stop_reason = reason;
    });
}
int32_t relp_get_b(relp_handle* h, double* out) {
    REQUIRE_LOADED(h);
    if (!out) return RELP_ERR_ARGUMENT;
    return guarded(h, [&] { h->solver->get_b(out); });
}
int32_t relp_get_objective(relp_handle* h, double* objective) {
    REQUIRE_LOADED(h);
    if (!objective) return RELP_ERR_ARGUMENT;
    return guarded(h, [&] { *objective = h->solver->objective(); });
}
int32_t relp_get_stats(const relp_handle* h, relp_stats* stats) {
    if (!h || !h->solver || !stats) return RELP_ERR_ARGUMENT;
    *stats = h->solver->stats();
    return RELP_OK;
}
int32_t relp_profile_kernel(relp_handle* h, int32_t which, int32_t repetitions, double* seconds) {
    REQUIRE_LOADED(h);
    if (!seconds || repetitions < 1 || which < 0 || which > 2) return RELP_ERR_ARGUMENT;
    return guarded(h, [&] { *seconds = h->solver->profile_kernel(which, repetitions); });
}
int32_t relp_debug_exact_finish(int32_t device, int32_t limbs, int32_t count, const uint64_t* T, const int32_t* carry, const int32_t* words,
                                int32_t shift, int32_t flip, uint64_t* N_out, int32_t* bits_out) {
    if (!T || !carry || !words || !N_out || !bits_out || count < 1 || shift < 0) return RELP_ERR_ARGUMENT;
    if (limbs != 16 && limbs != 32 && limbs != 64 && limbs != 128) return RELP_ERR_ARGUMENT;
    try {
        relp::exact_finish_entries(device, limbs, count, (const unsigned long long*)T, carry, words, shift, flip, (unsigned long long*)N_out, bits_out);
        return RELP_OK;
    } catch (const std::exception&) {
        return RELP_ERR_DEVICE;
    }
}

int32_t relp_debug_exact_words(int32_t device, int32_t limbs, int32_t mode, int32_t count, const uint64_t* a, const uint64_t* b, uint64_t* out) {
    if (!a || !b || !out || count < 1 || mode < 0 || mode > 2) return RELP_ERR_ARGUMENT;
    if (limbs != 16 && limbs != 32 && limbs != 64 && limbs != 128) return RELP_ERR_ARGUMENT;
    try {
        relp::exact_words_test(device, limbs, mode, count, (const unsigned long long*)a, (const unsigned long long*)b, (unsigned long long*)out);
        return RELP_OK;
    } catch (const std::exception&) {
        return RELP_ERR_DEVICE;
    }
}

int32_t relp_debug_grid_barrier(int32_t device, int32_t grid, int32_t rounds, int32_t reads, int32_t mode, int64_t limit_ticks, int64_t* out8) {
    if (!out8 || grid < 1 || rounds < 1 || reads < 0 || mode < 0 || mode > 1 || limit_ticks < 0) return RELP_ERR_ARGUMENT;
    try {
        relp::grid_barrier_test(device, grid, rounds, reads, mode, (long long)limit_ticks, (long long*)out8);
        return RELP_OK;
    } catch (const std::invalid_argument&) {
        return RELP_ERR_ARGUMENT;
    } catch (const std::exception&) {
        return RELP_ERR_DEVICE;
    }
}

int32_t relp_debug_exact_tile_bench(int32_t device, int32_t limbs, int32_t tiles, int32_t blocks, int32_t terms, int32_t shift, double* seconds) {
    if (!seconds || tiles < 1 || blocks < 1 || blocks > limbs / 8 || terms < 1 || terms > 2 || shift < 0) return RELP_ERR_ARGUMENT;
    try {
        *seconds = relp::exact_tile_bench(device, limbs, tiles, blocks, terms, shift);
        return RELP_OK;
    } catch (const std::invalid_argument&) {
        return RELP_ERR_ARGUMENT;
    } catch (const std::exception&) {
        return RELP_ERR_DEVICE;
    }
}

int32_t relp_debug_set_tuning(const relp_options* options) {
    relp_options adopted;
    if (adopt_options(options, &adopted) != RELP_OK) return RELP_ERR_ARGUMENT;
    thread_tuning() = tuning_of(adopted);
    return RELP_OK;
}
int32_t relp_debug_stamps(relp_handle* h, uint64_t* out64) {
    REQUIRE_LOADED(h);
    if (!out64) return RELP_ERR_ARGUMENT;
    return guarded(h, [&] { h->solver->debug_stamps(reinterpret_cast<unsigned long long*>(out64)); });
}
int32_t relp_reset_stats(relp_handle* h) {
    if (!h || !h->solver) return RELP_ERR_ARGUMENT;
    h->solver->reset_stats();
    return RELP_OK;
}

}  // extern "C"

// MPS (fixed / free) reader and standardisation: the step immediately before the hot path.
//
// Replaces, for the un-presolved pipeline of tests/netlib/mod.rs:47-71 of the reference:
//   io/mps/parse/{mod,fixed,free}.rs   section reader (rows sorted by NAME: parse/mod.rs:243-262)
//   io/mps/number/parse.rs:77-119      exact decimal -> rational (no exponent syntax)
//   io/mps/convert.rs:29-394           bounds (GLPK-like UP rule :211-216), ranges, rhs -> GeneralForm
//   general_form/mod.rs:325-332        standardize(): split free, flip/shift to x>=0, b>=0, minimise, E|R|L|G order
//   general_form/mod.rs:262-304        derive_matrix_data()
//   general_form/mod.rs:335-463        presolve() (optional: presolve.hpp)
// `standardize_general_form` is the part after the MPS conversion: it is also the entry for a caller that builds the
// general form itself (GeneralForm::new, general_form/mod.rs:211-237; C ABI relp_model_from_general_form).
#include <algorithm>
#include <map>
#include <sstream>
#include <stdexcept>
#include <unordered_map>

#include "model.hpp"
#include "presolve.hpp"

namespace relp {
namespace {

Rat parse_number(const std::string& text) {
    if (text.empty()) throw std::runtime_error("empty number");
    size_t pos = 0;
    bool negative = false;
    if (text[0] == '-') { negative = true; pos = 1; }
    i128 integer = 0;
    int steps = 0;
    bool seen_dot = false;
    for (; pos < text.size(); ++pos) {
        char c = text[pos];
        if (c == '.') {
            if (seen_dot) throw std::runtime_error("bad number: " + text);
            seen_dot = true;
        } else if (c >= '0' && c <= '9') {
            integer = add_checked(mul_checked(integer, 10), c - '0');
            if (seen_dot) ++steps;
        } else {
            throw std::runtime_error("bad number (exponents are not accepted, number/parse.rs:77): " + text);
        }
    }
    i128 den = 1;
    for (int k = 0; k < steps; ++k) den = mul_checked(den, 10);
    return Rat(negative ? -integer : integer, den);
}

std::string trim(const std::string& s) {
    size_t a = s.find_first_not_of(" \t\r");
    if (a == std::string::npos) return "";
    size_t b = s.find_last_not_of(" \t\r");
    return s.substr(a, b - a + 1);
}

// io/mps/parse/fixed.rs:137-145
const int FIELD_START[7] = {0, 1, 4, 14, 24, 39, 49};
const int FIELD_END[7] = {1, 3, 12, 22, 36, 47, 61};

std::string field(const std::string& line, int k) {
    if ((int)line.size() <= FIELD_START[k]) return "";
    int end = std::min<int>(FIELD_END[k], (int)line.size());
    return trim(line.substr(FIELD_START[k], end - FIELD_START[k]));
}

std::vector<std::string> split_ws(const std::string& line) {
    std::istringstream in(line);
    std::vector<std::string> out;
    std::string tok;
    while (in >> tok) out.push_back(tok);
    return out;
}

struct Raw {
    std::string name;
    bool maximize = false;
    std::string cost_row;
    std::vector<std::pair<std::string, RowKind>> rows;  // sorted by name
    std::vector<std::string> column_names;
    std::vector<SparseColumn> columns;
    std::vector<Rat> cost;
    std::vector<std::vector<std::pair<int, Rat>>> rhs_groups, range_groups;
    struct Bound { int column; std::string type; Rat value; };
    std::vector<Bound> bounds;
};

Raw parse(const std::string& text, bool fixed) {
    Raw raw;
    std::vector<std::string> lines;
    {
        std::istringstream in(text);
        std::string line;
        while (std::getline(in, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (line.empty()) continue;
            std::string t = trim(line);
            if (!t.empty() && t[0] == '*') continue;  // parse/mod.rs:97-103
            lines.push_back(line);
        }
    }
    if (lines.empty() || lines[0].compare(0, 4, "NAME") != 0) throw std::runtime_error("expected NAME");
    {
        auto toks = split_ws(lines[0].substr(4));
        raw.name = toks.empty() ? "" : toks[0];
    }
    std::string section;
    std::vector<std::pair<std::string, RowKind>> rows_unsorted;
    std::unordered_map<std::string, int> row_index, column_index;
    std::vector<std::vector<std::pair<std::string, Rat>>> column_entries;
    std::string last_rhs_name, last_range_name;
    bool have_rhs_group = false, have_range_group = false;

    auto data_fields = [&](const std::string& line) {
        if (fixed) return std::vector<std::string>{field(line, 2), field(line, 3), field(line, 4), field(line, 5), field(line, 6)};
        return split_ws(line);
    };

    for (size_t li = 1; li < lines.size(); ++li) {
        const std::string& line = lines[li];
        if (line[0] != ' ' && line[0] != '\t') {
            section = split_ws(line)[0];
            if (section == "ENDATA") break;
            if (section == "COLUMNS") {
                raw.rows = rows_unsorted;
                std::sort(raw.rows.begin(), raw.rows.end(),
                          [](const auto& a, const auto& b) { return a.first < b.first; });
                for (size_t i = 0; i < raw.rows.size(); ++i) {
                    if (!row_index.emplace(raw.rows[i].first, (int)i).second) throw std::runtime_error("Duplicate row name");
                }
                if (raw.cost_row.empty()) throw std::runtime_error("No cost name read.");
                if (row_index.count(raw.cost_row)) throw std::runtime_error("Cost row name found in other rows.");
            }
            continue;
        }
        if (section == "OBJSENSE") {
            std::string w = trim(line);
            raw.maximize = (w == "MAXIMIZE" || w == "MAX");
        } else if (section == "ROWS") {
            std::string type, name;
            if (fixed) { type = field(line, 1); name = field(line, 2); }
            else { auto t = split_ws(line); type = t.at(0); name = t.at(1); }
            if (type == "N") {
                if (!raw.cost_row.empty()) throw std::runtime_error("Second cost row detected.");
                raw.cost_row = name;
            } else if (type == "E") rows_unsorted.push_back({name, EQUAL});
            else if (type == "L") rows_unsorted.push_back({name, LESS});
            else if (type == "G") rows_unsorted.push_back({name, GREATER});
            else throw std::runtime_error("unknown row type " + type);
        } else if (section == "COLUMNS") {
            if (line.find("'MARKER'") != std::string::npos) continue;
            auto f = data_fields(line);
            if (f.size() < 3) throw std::runtime_error("short COLUMNS line");
            auto it = column_index.find(f[0]);
            int j;
            if (it == column_index.end()) {
                j = (int)raw.column_names.size();
                column_index.emplace(f[0], j);
                raw.column_names.push_back(f[0]);
                column_entries.emplace_back();
            } else {
                j = it->second;
            }
            column_entries[j].push_back({f[1], parse_number(f[2])});
            if (f.size() >= 5 && !f[3].empty() && !f[4].empty()) column_entries[j].push_back({f[3], parse_number(f[4])});
        } else if (section == "RHS" || section == "RANGES") {
            bool is_rhs = section == "RHS";
            auto f = data_fields(line);
            if (!fixed && f.size() % 2 == 0) f.insert(f.begin(), "");
            auto& groups = is_rhs ? raw.rhs_groups : raw.range_groups;
            std::string& last = is_rhs ? last_rhs_name : last_range_name;
            bool& have = is_rhs ? have_rhs_group : have_range_group;
            if (!have || last != f[0]) { groups.emplace_back(); last = f[0]; have = true; }
            auto add = [&](const std::string& row, const std::string& value) {
                auto it = row_index.find(row);
                if (it == row_index.end()) throw std::runtime_error("Row \"" + row + "\" not known.");  // parse/mod.rs:608-610
                groups.back().push_back({it->second, parse_number(value)});
            };
            add(f.at(1), f.at(2));
            if (f.size() >= 5 && !f[3].empty() && !f[4].empty()) add(f[3], f[4]);
        } else if (section == "BOUNDS") {
            std::string type, column, value;
            if (fixed) { type = field(line, 1); column = field(line, 3); value = field(line, 4); }
            else {
                auto t = split_ws(line);
                type = t.at(0);
                bool valued = !(type == "FR" || type == "MI" || type == "PL" || type == "BV");
                size_t need = valued ? 4 : 3;
                size_t base = t.size() >= need ? 2 : 1;  // bound-set name may be omitted
                column = t.at(base);
                if (valued) value = t.at(base + 1);
            }
            auto it = column_index.find(column);
            if (it == column_index.end()) throw std::runtime_error("Column name \"" + column + "\" unknown");
            Raw::Bound bd{it->second, type, Rat(0)};
            if (type == "LO" || type == "UP" || type == "FX" || type == "LI" || type == "UI") bd.value = parse_number(value);
            raw.bounds.push_back(bd);
        } else {
            throw std::runtime_error("unexpected section " + section);
        }
    }

    raw.columns.resize(raw.column_names.size());
    raw.cost.assign(raw.column_names.size(), Rat(0));
    for (size_t j = 0; j < column_entries.size(); ++j) {
        std::vector<std::pair<int, Rat>> values;
        for (auto& [row, value] : column_entries[j]) {
            if (row == raw.cost_row) { raw.cost[j] = value; continue; }
            auto it = row_index.find(row);
            if (it == row_index.end()) throw std::runtime_error("Row \"" + row + "\" not known.");
            values.push_back({it->second, value});
        }
        std::sort(values.begin(), values.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
        for (size_t k = 0; k < values.size(); ++k) {
            if (k > 0 && values[k].first == values[k - 1].first) throw std::runtime_error("Duplicate row for column");
            if (!values[k].second.is_zero()) raw.columns[j].push(values[k].first, values[k].second);
        }
    }
    return raw;
}

void replace_if(bool& has, Rat& current, const Rat& value, bool keep_greater) {
    if (!has) { has = true; current = value; return; }
    if (keep_greater ? (value > current) : (value < current)) current = value;
}

}  // namespace

StandardForm load_mps(const std::string& text, bool fixed_format, bool presolve_first) {
    Raw raw = parse(text, fixed_format);
    int nr_rows = (int)raw.rows.size();
    int n = (int)raw.columns.size();

    // ---- convert.rs:118-262 bounds --------------------------------------------------------------
    std::vector<GeneralVariable> vars(n);
    for (int j = 0; j < n; ++j) vars[j].cost = raw.cost[j];
    std::vector<char> needs_default_lower(n, 1), is_free(n, 0);
    for (const auto& bd : raw.bounds) {
        GeneralVariable& v = vars[bd.column];
        bool needs_lower = false;
        const std::string& t = bd.type;
        if (t == "LO" || t == "LI") replace_if(v.has_lower, v.lower, bd.value, true);
        else if (t == "UP" || t == "UI") { replace_if(v.has_upper, v.upper, bd.value, false); needs_lower = true; }
        else if (t == "FX") { replace_if(v.has_lower, v.lower, bd.value, true); replace_if(v.has_upper, v.upper, bd.value, false); }
        else if (t == "FR") { if (v.has_lower || v.has_upper) throw std::runtime_error("Variable can't be bounded and free"); is_free[bd.column] = 1; }
        else if (t == "MI") replace_if(v.has_upper, v.upper, Rat(0), false);  // sic: convert.rs:233-237
        else if (t == "PL") replace_if(v.has_lower, v.lower, Rat(0), true);
        else if (t == "BV") { replace_if(v.has_lower, v.lower, Rat(0), true); replace_if(v.has_upper, v.upper, Rat(1), false); }
        else throw std::runtime_error("Bound type \"" + t + "\" unknown.");
        needs_default_lower[bd.column] = needs_default_lower[bd.column] && needs_lower;
    }
    for (int j = 0; j < n; ++j) {
        if (is_free[j] && (vars[j].has_lower || vars[j].has_upper)) throw std::runtime_error("A variable is both free and bounded.");
        if (needs_default_lower[j]) { vars[j].has_lower = true; vars[j].lower = Rat(0); }
    }

    // ---- convert.rs:264-394 ranges, constraint types, b -----------------------------------------
    std::vector<RowKind> kind(nr_rows);
    std::vector<Rat> range(nr_rows);
    std::vector<char> has_range(nr_rows, 0);
    for (auto& group : raw.range_groups)
        for (auto& [i, r] : group) {
            if (has_range[i]) throw std::runtime_error("Only one range per row can be specified.");
            has_range[i] = 1;
            range[i] = r;
        }
    for (int i = 0; i < nr_rows; ++i) {
        if (has_range[i]) kind[i] = range[i].is_zero() ? EQUAL : RANGE;
        else kind[i] = raw.rows[i].second;
    }
    std::vector<Rat> b(nr_rows);
    std::vector<char> has_b(nr_rows, 0);
    for (auto& group : raw.rhs_groups) {
        std::vector<std::pair<int, Rat>> sorted = group;
        std::stable_sort(sorted.begin(), sorted.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
        for (auto& [i, value] : sorted) {
            RowKind original = raw.rows[i].second;
            if (!has_b[i]) {
                has_b[i] = 1;
                if (kind[i] == RANGE) {
                    int sign = range[i].sign();
                    if (sign < 0) range[i] = -range[i];
                    if (original == GREATER) b[i] = value + range[i];
                    else if (original == LESS) b[i] = value;
                    else b[i] = sign >= 0 ? value + range[i] : value;
                } else {
                    b[i] = value;
                }
            } else {
                if (original == EQUAL) { if (value != b[i]) throw std::runtime_error("Trivial infeasibility"); }
                else if (original == GREATER) { if (value > b[i]) b[i] = value; }
                else { if (value < b[i]) b[i] = value; }
            }
        }
    }

    GeneralInput general;
    general.name = raw.name;
    general.maximize = raw.maximize;
    general.variables = std::move(vars);
    general.columns = std::move(raw.columns);
    general.kind = std::move(kind);
    general.range = std::move(range);
    general.b = std::move(b);
    general.column_names = std::move(raw.column_names);
    return standardize_general_form(std::move(general), presolve_first);
}

// presolve_level: 0 none | 1 the reference's presolve as it is | 2, 3 the same without the implied bounds that need more than 126 / 60
// bits.  Throws RatOverflow when a value leaves the 128-bit rationals of the host model anywhere on the way.
static StandardForm standardize_at_level(GeneralInput general, int presolve_level) {
    const bool presolve_first = presolve_level > 0;
    std::vector<GeneralVariable>& vars = general.variables;
    std::vector<RowKind>& kind = general.kind;
    std::vector<Rat>& range = general.range;
    std::vector<Rat>& b = general.b;
    int nr_rows = (int)b.size();
    int n = (int)vars.size();
    if ((int)general.columns.size() != n || (int)kind.size() != nr_rows || (int)range.size() != nr_rows ||
        (int)general.column_names.size() != n)
        throw std::runtime_error("general form: inconsistent dimensions");
    for (const auto& column : general.columns)
        for (size_t k = 0; k < column.nnz(); ++k) {
            if (column.index[k] < 0 || column.index[k] >= nr_rows) throw std::runtime_error("general form: row index out of range");
            if (k > 0 && column.index[k] <= column.index[k - 1]) throw std::runtime_error("general form: column entries must ascend by row");
            if (column.value[k].is_zero()) throw std::runtime_error("general form: explicit zero in a sparse column");
        }
    // ---- general_form/mod.rs:335-463 presolve (optional; the reference's harness applies it: tests/netlib/mod.rs:58) ----
    StandardForm out;
    out.name = general.name;
    out.all_column_names = general.column_names;
    std::vector<SparseColumn> columns = std::move(general.columns);
    Rat fixed_cost = general.fixed_cost;
    {
        if (presolve_first) {
            // The presolve in arbitrary precision (presolve.hpp), then back to the 128-bit rationals of the host model -- everything
            // is converted BEFORE anything is committed.  (The levels: standardize_general_form below.)
            auto attempt = [&](size_t bit_limit) -> bool {
                GeneralProblem gp;
                gp.maximize = general.maximize;
                gp.fixed_cost = Num(general.fixed_cost);
                for (int j = 0; j < n; ++j) {
                    PVariable v;
                    v.cost = Num(vars[j].cost);
                    v.has_lower = vars[j].has_lower;
                    v.has_upper = vars[j].has_upper;
                    v.lower = Num(vars[j].lower);
                    v.upper = Num(vars[j].upper);
                    gp.variables.push_back(v);
                    PColumn column;
                    for (size_t k = 0; k < columns[j].nnz(); ++k) column.push(columns[j].index[k], Num(columns[j].value[k]));
                    gp.columns.push_back(column);
                    gp.active_to_original.push_back(j);
                }
                for (int i = 0; i < nr_rows; ++i) {
                    gp.b.push_back(Num(b[i]));
                    ConstraintKind ck;
                    ck.kind = kind[i];
                    if (kind[i] == RANGE) ck.range = Num(range[i]);
                    gp.kinds.push_back(ck);
                }
                presolve(gp, bit_limit);
                if (gp.variables.empty() || gp.b.empty())
                    throw std::runtime_error("presolve: the problem was solved completely (no rows or columns remain)");
                if (diagnostic("RELP_PRESOLVE_DEBUG")) {  // diagnostic: which quantities of the presolved LP do not fit 128-bit rationals
                    int bounds = 0, coefficients = 0, rhs = 0, ranges = 0, fixed = 0, removed_n = 0;
                    size_t widest = 0;
                    auto fits = [&](const BigRat& v, int& counter) {
                        try {
                            (void)v.to_rat();
                        } catch (const RatOverflow&) {
                            ++counter;
                            widest = std::max(widest, v.bits());
                        }
                    };
                    for (size_t j = 0; j < gp.variables.size(); ++j) {
                        if (gp.variables[j].has_lower) fits(gp.variables[j].lower, bounds);
                        if (gp.variables[j].has_upper) fits(gp.variables[j].upper, bounds);
                        for (size_t k = 0; k < gp.columns[j].nnz(); ++k) fits(gp.columns[j].value[k], coefficients);
                    }
                    for (size_t i = 0; i < gp.b.size(); ++i) {
                        fits(gp.b[i], rhs);
                        if (gp.kinds[i].kind == RANGE) fits(gp.kinds[i].range, ranges);
                    }
                    fits(gp.fixed_cost, fixed);
                    for (auto& [original, how] : gp.removed) {
                        fits(how.constant, removed_n);
                        for (auto& [k, c] : how.coefficients) fits(c, removed_n);
                    }
                    fprintf(stderr, "[presolve] bit limit %zu: %zu x %zu left; not representable: bounds %d, coefficients %d, rhs %d, ranges %d, fixed cost %d, "
                                    "removed-variable records %d (widest %zu bits)\n",
                            bit_limit, gp.b.size(), gp.variables.size(), bounds, coefficients, rhs, ranges, fixed, removed_n, widest);
                }
                try {
                    std::vector<GeneralVariable> kept;
                    std::vector<SparseColumn> kept_columns;
                    for (size_t j = 0; j < gp.variables.size(); ++j) {
                        GeneralVariable v = vars[gp.active_to_original[j]];  // cost, shift, flipped are untouched by the presolve
                        v.has_lower = gp.variables[j].has_lower;
                        v.has_upper = gp.variables[j].has_upper;
                        v.lower = v.has_lower ? gp.variables[j].lower.to_rat() : Rat(0);
                        v.upper = v.has_upper ? gp.variables[j].upper.to_rat() : Rat(0);
                        kept.push_back(v);
                        SparseColumn column;
                        for (size_t k = 0; k < gp.columns[j].nnz(); ++k) column.push(gp.columns[j].index[k], gp.columns[j].value[k].to_rat());
                        kept_columns.push_back(column);
                    }
                    std::vector<Rat> new_b, new_range(gp.b.size(), Rat(0));
                    std::vector<RowKind> new_kind;
                    for (size_t i = 0; i < gp.b.size(); ++i) {
                        new_b.push_back(gp.b[i].to_rat());
                        new_kind.push_back(gp.kinds[i].kind);
                        if (gp.kinds[i].kind == RANGE) new_range[i] = gp.kinds[i].range.to_rat();
                    }
                    const Rat new_fixed_cost = gp.fixed_cost.to_rat();
                    std::vector<std::pair<int, RemovedOriginal>> removed;
                    for (auto& [original, how] : gp.removed) {
                        RemovedOriginal r;
                        r.function_of_others = how.function_of_others;
                        r.constant = how.constant.to_rat();
                        for (auto& [k, c] : how.coefficients) r.coefficients.push_back({k, c.to_rat()});
                        removed.push_back({original, r});
                    }
                    vars.swap(kept);
                    columns.swap(kept_columns);
                    b.swap(new_b);
                    kind.swap(new_kind);
                    range.swap(new_range);
                    fixed_cost = new_fixed_cost;
                    out.active_to_original = gp.active_to_original;
                    out.removed = std::move(removed);
                    return true;
                } catch (const RatOverflow&) {
                    return false;
                }
            };
            const size_t limits[] = {0, 0, 126, 60};
            if (!attempt(limits[presolve_level])) throw RatOverflow();
            out.presolve_state = presolve_level == 1 ? StandardForm::PRESOLVE_APPLIED : StandardForm::PRESOLVE_BOUNDED;
        } else {
            for (int j = 0; j < n; ++j) out.active_to_original.push_back(j);
        }
    }
    nr_rows = (int)b.size();
    n = (int)vars.size();
    // ---- general_form/mod.rs:506-587 transform_variables ----------------------------------------
    out.nr_original = n;
    for (int j = 0; j < n; ++j) out.column_names.push_back(out.all_column_names[out.active_to_original[j]]);
    out.free_negative_part.assign(n, -1);
    for (int j = 0; j < n; ++j) {
        if (!vars[j].has_lower && !vars[j].has_upper) {
            out.free_negative_part[j] = (int)columns.size();
            SparseColumn neg = columns[j];
            for (auto& v : neg.value) v = -v;
            columns.push_back(neg);
            GeneralVariable twin;
            twin.cost = -vars[j].cost;
            twin.has_lower = true;
            vars.push_back(twin);
            vars[j].has_lower = true;
            vars[j].lower = Rat(0);
        }
    }
    for (size_t j = 0; j < vars.size(); ++j) {
        GeneralVariable& v = vars[j];
        if (!v.has_lower && v.has_upper) {
            v.flipped = !v.flipped;
            v.shift = -v.shift;
            v.cost = -v.cost;
            v.has_lower = true;
            v.lower = -v.upper;
            v.has_upper = false;
            for (auto& c : columns[j].value) c = -c;
        }
        if (v.has_lower) {
            v.shift = v.shift - v.lower;
            if (v.has_upper) v.upper = v.upper - v.lower;
            fixed_cost = fixed_cost + v.lower * v.cost;
            if (!v.lower.is_zero())
                for (size_t k = 0; k < columns[j].nnz(); ++k) b[columns[j].index[k]] = b[columns[j].index[k]] - columns[j].value[k] * v.lower;
            v.lower = Rat(0);
        }
    }
    // ---- general_form/mod.rs:592-618 make_b_non_negative -----------------------------------------
    std::vector<char> negate(nr_rows, 0);
    for (int i = 0; i < nr_rows; ++i) negate[i] = b[i].sign() < 0;
    for (auto& column : columns)
        for (size_t k = 0; k < column.nnz(); ++k)
            if (negate[column.index[k]]) column.value[k] = -column.value[k];
    for (int i = 0; i < nr_rows; ++i) {
        if (!negate[i]) continue;
        if (kind[i] == LESS) { kind[i] = GREATER; b[i] = -b[i]; }
        else if (kind[i] == GREATER) { kind[i] = LESS; b[i] = -b[i]; }
        else if (kind[i] == EQUAL) b[i] = -b[i];
        else b[i] = range[i] - b[i];
    }
    // ---- general_form/mod.rs:623-633 ----------------------------------------------------------------
    if (general.maximize)
        for (auto& v : vars) v.cost = -v.cost;
    // ---- general_form/mod.rs:651-717 reorder_constraints_by_type (stable) ------------------------
    std::vector<int> order(nr_rows);
    for (int i = 0; i < nr_rows; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return (int)kind[x] < (int)kind[y]; });
    std::vector<int> destination(nr_rows);
    for (int d = 0; d < nr_rows; ++d) destination[order[d]] = d;
    MatrixData& data = out.data;
    int counts[4] = {0, 0, 0, 0};
    for (int i = 0; i < nr_rows; ++i) counts[kind[i]]++;
    data.nr_equality = counts[EQUAL];
    data.nr_range = counts[RANGE];
    data.nr_upper = counts[LESS];
    data.nr_lower = counts[GREATER];
    data.b.resize(nr_rows);
    for (int i = 0; i < nr_rows; ++i) data.b[destination[i]] = b[i];
    for (int d = 0; d < nr_rows; ++d)
        if (kind[order[d]] == RANGE) data.ranges.push_back(range[order[d]]);
    data.constraints.resize(columns.size());
    for (size_t j = 0; j < columns.size(); ++j) {
        std::vector<std::pair<int, Rat>> entries;
        for (size_t k = 0; k < columns[j].nnz(); ++k) entries.push_back({destination[columns[j].index[k]], columns[j].value[k]});
        std::sort(entries.begin(), entries.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
        for (auto& [i, v] : entries) data.constraints[j].push(i, v);
    }
    data.variables.resize(vars.size());
    for (size_t j = 0; j < vars.size(); ++j) {
        data.variables[j].cost = vars[j].cost;
        data.variables[j].has_upper = vars[j].has_upper;
        data.variables[j].upper = vars[j].upper;
        data.variables[j].shift = vars[j].shift;
        data.variables[j].flipped = vars[j].flipped;
    }
    data.finalize();
    out.fixed_cost = fixed_cost;
    return out;
}

// `standardize()` of the reference's pipeline, with its presolve in front when asked for.  The OVERFLOW escalation of the host
// model: the presolve computes in arbitrary precision, the model is 128-bit rationals.  Level 1 is the
// reference's presolve as it is.  When a value does not fit anywhere between the presolve and the standard form -- BORE3D, CYCLE,
// GREENBEB: a handful of variable bounds that domain propagation tightens to hundreds or thousands of bits, and what the shifts
// make of them -- the presolve runs again WITHOUT the implied bounds that need more than 126 bits, then more than 60 (valid,
// slightly weaker reductions: an implied bound of a constraint that stays in the problem can always be left out); only when
// those fail too is the LP loaded as the file states it.  The optimum is the same at every level.
StandardForm standardize_general_form(GeneralInput general, bool presolve_first) {
    if (!presolve_first) return standardize_at_level(std::move(general), 0);
    for (int level = 1; level <= 3; ++level) {
        try {
            return standardize_at_level(general, level);  // (a copy: the next level starts from the same input)
        } catch (const RatOverflow&) {
        }
    }
    StandardForm out = standardize_at_level(std::move(general), 0);
    out.presolve_state = StandardForm::PRESOLVE_DROPPED;
    out.presolve_dropped = true;
    return out;
}

}  // namespace relp

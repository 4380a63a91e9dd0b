"""MPS reader and ``GeneralForm`` standardisation (oracle; test infrastructure only).

Follows ``/root/reference/src/io/mps/{parse/mod.rs, parse/fixed.rs, parse/free.rs, number/parse.rs,
convert.rs}`` and ``/root/reference/src/data/linear_program/general_form/mod.rs`` (``standardize``,
``derive_matrix_data``, ``compute_full_solution_with_reduced_solution``).  The reference's presolve
(``general_form/presolve/**``) is NOT restated: the oracle and the
product both solve the un-presolved standard form, whose optimum is identical.
"""
from fractions import Fraction

from .provider import MatrixData, Variable

ZERO = Fraction(0)

# io/mps/parse/fixed.rs:137-145
FIELDS = [(0, 1), (1, 3), (4, 12), (14, 22), (24, 36), (39, 47), (49, 61)]


def parse_number(text):
    """io/mps/number/parse.rs:77-119: ``[-]int[.frac]`` -> exact rational (no exponent syntax)."""
    negative = text.startswith("-")
    if negative:
        text = text[1:]
    elif text.startswith("+"):
        raise ValueError("leading '+' is not accepted by the reference parser: %r" % text)
    if "." in text:
        integer_part, mantissa = text.split(".", 1)
        steps = len(mantissa)
        integer = int(integer_part or "0") * 10 ** steps + int(mantissa or "0")
    else:
        steps = 0
        integer = int(text)
    value = Fraction(integer, 10 ** steps)
    return -value if negative else value


class MPS:
    """io/mps/mod.rs (struct MPS)."""

    def __init__(self):
        self.name = ""
        self.objective = "Minimize"
        self.cost_row_name = None
        self.rows = []          # [(name, type)] sorted by name (parse/mod.rs:243-262)
        self.columns = []       # [(name, [(row, value)])]
        self.cost_values = []   # [(column, value)]
        self.rhss = []          # [(name, [(row, value)])]
        self.ranges = []
        self.bounds = []        # [(name, [(column, type, value)])]


def _fields_fixed(line, which):
    out = []
    for k in which:
        a, b = FIELDS[k]
        out.append(line[a:b].strip())
    return out


def parse(text, fixed=True):
    """io/mps/parse/mod.rs:40-95.  ``fixed`` selects byte-column splitting (parse/fixed.rs) over
    whitespace splitting (parse/free.rs)."""
    lines = [ln for ln in text.splitlines() if ln and not ln.lstrip().startswith("*")]
    mps = MPS()
    it = iter(lines)
    first = next(it)
    assert first.startswith("NAME"), first
    rest = first[4:].split()
    mps.name = rest[0] if rest else ""

    section = None
    rows_unsorted = []
    row_index = None
    column_data = {}      # name -> list[(row name, value)]
    column_order = []
    rhs_groups, range_groups, bound_groups = [], [], []

    def split(line, n_lead):
        """Fields 2..6 of a data line (names and numbers)."""
        if fixed:
            parts = _fields_fixed(line, range(2, 7))
            return [p for p in parts]
        return line.split()

    pending = None
    for line in it:
        if not line.startswith(" "):
            word = line.split()[0]
            if word == "OBJSENSE":
                pending = "OBJSENSE"
                section = "OBJSENSE"
                continue
            if word == "ENDATA":
                section = "ENDATA"
                break
            section = word
            if section == "COLUMNS":
                # parse/mod.rs:243-262: rows are sorted by name before indexing
                mps.rows = sorted(rows_unsorted, key=lambda r: r[0])
                names = [r[0] for r in mps.rows]
                assert len(set(names)) == len(names), "Duplicate row name"
                assert mps.cost_row_name is not None, "No cost name read."
                assert mps.cost_row_name not in set(names)
                row_index = {name: i for i, name in enumerate(names)}
            continue
        if section == "OBJSENSE":
            word = line.strip()
            mps.objective = "Maximize" if word in ("MAXIMIZE", "MAX") else "Minimize"
        elif section == "ROWS":
            if fixed:
                row_type, name = _fields_fixed(line, (1, 2))
            else:
                row_type, name = line.split()[:2]
            if row_type == "N":
                assert mps.cost_row_name is None, "Second cost row detected."
                mps.cost_row_name = name
            else:
                rows_unsorted.append((name, {"E": "Equal", "L": "Less", "G": "Greater"}[row_type]))
        elif section == "COLUMNS":
            if "'MARKER'" in line:
                continue  # integrality markers: irrelevant for the relaxation
            parts = split(line, 1)
            name = parts[0]
            if name not in column_data:
                column_data[name] = []
                column_order.append(name)
            pairs = [(parts[1], parts[2])]
            if len(parts) >= 5 and parts[3] and parts[4]:
                pairs.append((parts[3], parts[4]))
            for row_name, value_text in pairs:
                column_data[name].append((row_name, parse_number(value_text)))
        elif section in ("RHS", "RANGES"):
            groups = rhs_groups if section == "RHS" else range_groups
            parts = split(line, 1)
            if not fixed and len(parts) % 2 == 0:
                parts = [""] + parts  # free format allows the set name to be omitted
            name = parts[0]
            if not groups or groups[-1][0] != name:
                groups.append((name, []))
            pairs = [(parts[1], parts[2])]
            if len(parts) >= 5 and parts[3] and parts[4]:
                pairs.append((parts[3], parts[4]))
            for row_name, value_text in pairs:
                if row_name not in row_index:
                    raise ValueError('Row "%s" not known.' % row_name)  # parse/mod.rs:608-610 (e.g. GROW7)
                groups[-1][1].append((row_index[row_name], parse_number(value_text)))
        elif section == "BOUNDS":
            if fixed:
                bound_type, bound_name, column_name, value_text = _fields_fixed(line, (1, 2, 3, 4))
            else:
                parts = line.split()
                bound_type = parts[0]
                if bound_type in ("FR", "MI", "PL", "BV"):
                    bound_name, column_name = (parts[1], parts[2]) if len(parts) >= 3 else ("", parts[1])
                    value_text = ""
                else:
                    bound_name, column_name, value_text = (parts[1], parts[2], parts[3]) if len(parts) >= 4 \
                        else ("", parts[1], parts[2])
            if not bound_groups or bound_groups[-1][0] != bound_name:
                bound_groups.append((bound_name, []))
            value = parse_number(value_text) if bound_type in ("LO", "UP", "FX", "LI", "UI") else None
            bound_groups[-1][1].append((column_name, bound_type, value))
        else:
            raise ValueError("Unexpected section %r" % section)

    column_index = {name: j for j, name in enumerate(column_order)}
    for j, name in enumerate(column_order):
        values = []
        for row_name, value in column_data[name]:
            if row_name == mps.cost_row_name:
                mps.cost_values.append((j, value))
            elif row_name in row_index:
                values.append((row_index[row_name], value))
            else:
                raise ValueError('Row "%s" not known.' % row_name)
        values.sort(key=lambda t: t[0])  # parse/mod.rs:372-373
        assert all(a[0] != b[0] for a, b in zip(values, values[1:])), "Duplicate row for column"
        mps.columns.append((name, values))
    for name, values in rhs_groups:
        mps.rhss.append((name, sorted(values, key=lambda t: t[0])))
    for name, values in range_groups:
        mps.ranges.append((name, sorted(values, key=lambda t: t[0])))
    for name, values in bound_groups:
        mps.bounds.append((name, [(column_index[c], t, v) for c, t, v in values]))
    return mps


class GeneralForm:
    """general_form/mod.rs:41-81 (only what ``standardize``/``derive_matrix_data`` need)."""

    def __init__(self, objective, columns, constraint_types, b, variables, names):
        self.objective = objective
        self.fixed_cost = ZERO
        self.columns = [list(c) for c in columns]        # column major
        self.constraint_types = list(constraint_types)   # "Equal" | "Less" | "Greater" | ("Range", r)
        self.b = list(b)
        self.variables = list(variables)
        self.names = list(names)
        self.nr_original = len(self.variables)
        self.free_pairs = {}                             # positive column -> negative column
        self.active_to_original = list(range(len(self.variables)))  # general_form/mod.rs `from_active_to_original`
        self.removed = {}                                # original index -> RemovedVariable (filled by presolve)

    # ---- construction from MPS (io/mps/convert.rs:29-90) -------------------------------------------
    @classmethod
    def from_mps(cls, mps):
        nr_rows = len(mps.rows)
        costs = dict(mps.cost_values)
        variables = []
        for j, (name, _) in enumerate(mps.columns):
            variables.append(Variable(costs.get(j, ZERO), lower_bound=None, upper_bound=None))
        _process_bounds(variables, mps.bounds)
        columns = [[(i, v) for i, v in values if v != 0] for _, values in mps.columns]

        # convert.rs:268-330: ranges (one per row), then constraint types
        range_rows = sorted((t for _, values in mps.ranges for t in values), key=lambda t: t[0])
        assert all(a[0] != b[0] for a, b in zip(range_rows, range_rows[1:])), "Only one range per row"
        ranges = dict(range_rows)
        constraint_types = []
        for i, (_, row_type) in enumerate(mps.rows):
            if i in ranges:
                constraint_types.append("Equal" if ranges[i] == 0 else ("Range", ranges[i]))
            else:
                constraint_types.append(row_type)

        # convert.rs:332-394 compute_b
        b = [None] * nr_rows
        for _, values in mps.rhss:
            for i, value in values:
                row_type = mps.rows[i][1]
                if b[i] is None:
                    if isinstance(constraint_types[i], tuple):
                        r = constraint_types[i][1]
                        sign = (r > 0) - (r < 0)
                        r = abs(r)
                        constraint_types[i] = ("Range", r)
                        if row_type == "Greater":
                            b[i] = value + r
                        elif row_type == "Less":
                            b[i] = value
                        else:
                            b[i] = value + r if sign >= 0 else value
                    else:
                        b[i] = value
                else:
                    if row_type == "Equal":
                        assert value == b[i], "Trivial infeasibility"
                    elif row_type == "Greater":
                        b[i] = max(b[i], value)
                    else:
                        b[i] = min(b[i], value)
        b = [ZERO if v is None else v for v in b]
        return cls(mps.objective, columns, constraint_types, b, variables, [n for n, _ in mps.columns])

    # ---- standardize (general_form/mod.rs:325-332) -------------------------------------------------
    def standardize(self):
        self._transform_variables()
        self._make_b_non_negative()
        if self.objective == "Maximize":  # mod.rs:623-633
            self.objective = "Minimize"
            for v in self.variables:
                v.cost = -v.cost
        return self._reorder_constraints_by_type()

    def _transform_variables(self):
        """mod.rs:506-548 with split_free_variables :554-587."""
        free = [j for j, v in enumerate(self.variables) if v.lower_bound is None and v.upper_bound is None]
        for j in free:
            self.free_pairs[j] = len(self.columns)
            self.columns.append([(i, -v) for i, v in self.columns[j]])
            self.variables.append(Variable(-self.variables[j].cost, lower_bound=ZERO))
            self.variables[j].lower_bound = ZERO
        for j, variable in enumerate(self.variables):
            if variable.lower_bound is None and variable.upper_bound is not None:
                variable.flipped = not variable.flipped
                variable.shift = -variable.shift
                variable.cost = -variable.cost
                variable.lower_bound = -variable.upper_bound
                variable.upper_bound = None
                self.columns[j] = [(i, -v) for i, v in self.columns[j]]
            if variable.lower_bound is not None:
                lower = variable.lower_bound
                variable.shift -= lower
                if variable.upper_bound is not None:
                    variable.upper_bound -= lower
                self.fixed_cost += lower * variable.cost
                for i, coefficient in self.columns[j]:
                    self.b[i] -= coefficient * lower
                variable.lower_bound = ZERO

    def _make_b_non_negative(self):
        """mod.rs:592-618."""
        negate = {i for i, v in enumerate(self.b) if v < 0}
        for j, column in enumerate(self.columns):
            self.columns[j] = [(i, -v if i in negate else v) for i, v in column]
        for i in sorted(negate):
            kind = self.constraint_types[i]
            if kind == "Less":
                self.constraint_types[i] = "Greater"
                self.b[i] = -self.b[i]
            elif kind == "Equal":
                self.b[i] = -self.b[i]
            elif kind == "Greater":
                self.constraint_types[i] = "Less"
                self.b[i] = -self.b[i]
            else:
                self.b[i] = kind[1] - self.b[i]

    def _reorder_constraints_by_type(self):
        """mod.rs:651-717: stable partition into E | R | L | G."""
        def group(kind):
            return {"Equal": 0, "Less": 2, "Greater": 3}.get(kind, 1) if not isinstance(kind, tuple) else 1
        order = sorted(range(len(self.b)), key=lambda i: group(self.constraint_types[i]))
        destination = {source: dest for dest, source in enumerate(order)}
        counts = [0, 0, 0, 0]
        for kind in self.constraint_types:
            counts[group(kind)] += 1
        self.b = [self.b[i] for i in order]
        self.constraint_types = [self.constraint_types[i] for i in order]
        self.columns = [sorted((destination[i], v) for i, v in column) for column in self.columns]
        return counts

    def derive_matrix_data(self, counts):
        """mod.rs:262-304."""
        nr_e, nr_r, nr_l, nr_g = counts
        ranges = [kind[1] for kind in self.constraint_types[nr_e:nr_e + nr_r]]
        return MatrixData(self.columns, self.b, ranges, nr_e, nr_r, nr_l, nr_g, self.variables)

    def objective_of(self, reduced_solution):
        """mod.rs:840-851: objective = sum_j x_j c_j (standardised) + fixed cost."""
        return sum((v * self.variables[j].cost for j, v in reduced_solution), ZERO) + self.fixed_cost

    def full_solution(self, reduced_solution):
        """mod.rs:840-934 without presolve: un-shift, un-flip, recombine free variables."""
        values = dict(reduced_solution)
        by_original = {}
        for j, original in enumerate(self.active_to_original):
            variable = self.variables[j]
            x = values.get(j, ZERO)
            if j in self.free_pairs:
                x = x - values.get(self.free_pairs[j], ZERO)
            # reshift_solution (mod.rs:753-771): value = (x - shift) flipped back
            x = x - variable.shift
            if variable.flipped:
                x = -x
            by_original[original] = x

        def resolve(original):  # RemovedVariable (mod.rs:96-118): solved, or an affine function of other variables
            if original not in by_original:
                solution = self.removed[original]
                if solution[0] == "Solved":
                    by_original[original] = solution[1]
                else:
                    _, constant, coefficients = solution
                    by_original[original] = constant - sum((c * resolve(k) for k, c in coefficients), ZERO)
            return by_original[original]

        return {self.names[j]: resolve(j) for j in range(self.nr_original)}


def _process_bounds(variables, bounds):
    """io/mps/convert.rs:118-262."""
    needs_default_lower = [True] * len(variables)
    is_free = [False] * len(variables)

    def replace(current, new, greater):
        if current is None:
            return new
        if greater:
            return new if new > current else current
        return new if new < current else current

    for _, values in bounds:
        for j, bound_type, value in values:
            v = variables[j]
            needs_lower = False
            if bound_type in ("LO", "LI"):
                v.lower_bound = replace(v.lower_bound, value, True)
            elif bound_type in ("UP", "UI"):
                v.upper_bound = replace(v.upper_bound, value, False)
                needs_lower = True
            elif bound_type == "FX":
                v.lower_bound = replace(v.lower_bound, value, True)
                v.upper_bound = replace(v.upper_bound, value, False)
            elif bound_type == "FR":
                assert v.lower_bound is None and v.upper_bound is None, "Variable can't be bounded and free"
                is_free[j] = True
            elif bound_type == "MI":
                v.upper_bound = replace(v.upper_bound, ZERO, False)  # sic: convert.rs:233-237
            elif bound_type == "PL":
                v.lower_bound = replace(v.lower_bound, ZERO, True)
            elif bound_type == "BV":
                v.lower_bound = replace(v.lower_bound, ZERO, True)
                v.upper_bound = replace(v.upper_bound, Fraction(1), False)
            else:
                raise ValueError("Bound type %r unknown." % bound_type)
            needs_default_lower[j] = needs_default_lower[j] and needs_lower
    for j, v in enumerate(variables):
        assert not (is_free[j] and (v.lower_bound is not None or v.upper_bound is not None)), \
            "A variable is both free and bounded."
        if needs_default_lower[j]:
            v.lower_bound = ZERO  # convert.rs:244-262


def load_problem(path, fixed=None, presolve=False):
    """Read an MPS/SIF file -> ``(GeneralForm (standardised), MatrixData)``.

    ``presolve=True`` applies ``GeneralForm::presolve`` first, as the reference's Netlib harness does
    (tests/netlib/mod.rs:58).

    The Netlib harness uses ``parse_fixed`` (tests/netlib/mod.rs:55); ``io::import`` uses the free
    parser for ``.mps`` (io/mod.rs:46).
    """
    with open(path) as handle:
        text = handle.read()
    if fixed is None:
        fixed = str(path).upper().endswith(".SIF")
    general = GeneralForm.from_mps(parse(text, fixed=fixed))
    if presolve:
        from .presolve import presolve as run_presolve
        run_presolve(general)
    counts = general.standardize()
    return general, general.derive_matrix_data(counts)

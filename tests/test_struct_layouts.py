"""The three hand-written mirrors of the C structs -- the `#[repr(C)]` structs of INTEGRATION.md (what a maintainer of the
reference would paste into the Rust shim), the ctypes classes of relp_amd/api.py and include/relp_amd.h itself -- must agree in
field order, names and widths (CPU; no library needed)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "relp_amd.h")).read()
INTEGRATION = open(os.path.join(ROOT, "INTEGRATION.md")).read()

C_WIDTH = {"int32_t": ("i", 4), "uint32_t": ("u", 4), "int64_t": ("i", 8), "double": ("f", 8), "uint8_t": ("u", 1)}
RUST_WIDTH = {"i32": ("i", 4), "u32": ("u", 4), "i64": ("i", 8), "c_double": ("f", 8), "f64": ("f", 8), "c_int": ("i", 4), "u8": ("u", 1)}


def strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def c_struct(name):
    """[(field, kind, bytes, count)] of `typedef struct name { ... } name;`, scalar and array fields only."""
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), HEADER, flags=re.S).group(1)
    fields = []
    for declaration in strip_c_comments(body).split(";"):
        declaration = " ".join(declaration.split())
        if not declaration:
            continue
        type_name, _, rest = declaration.partition(" ")
        declarators = [d.strip() for d in rest.split(",")]
        if type_name not in C_WIDTH or not all(re.fullmatch(r"\w+(\[\d+\])?", d) for d in declarators):
            fields.append((declaration, "other", 0, 1))  # pointers, callbacks, nested structs: compared by name only
            continue
        kind, width = C_WIDTH[type_name]
        for d in declarators:  # `int32_t worker, device;`
            match = re.fullmatch(r"(\w+)(?:\[(\d+)\])?", d)
            fields.append((match.group(1), kind, width, int(match.group(2) or 1)))
    return fields


def rust_struct(name):
    body = re.search(r"pub struct %s \{(.*?)\n?\}" % name, INTEGRATION, flags=re.S).group(1)
    body = re.sub(r"//[^\n]*", "", body)
    fields = []
    for declaration in body.split(","):
        declaration = " ".join(declaration.split())
        if not declaration:
            continue
        match = re.fullmatch(r"pub (\w+): (?:\[(\w+); (\d+)\]|(\w+))", declaration)
        assert match, declaration
        type_name = match.group(2) or match.group(4)
        kind, width = RUST_WIDTH[type_name]
        fields.append((match.group(1), kind, width, int(match.group(3) or 1)))
    return fields


def ctypes_struct(cls):
    kinds = {C.c_int32: ("i", 4), C.c_uint32: ("u", 4), C.c_int64: ("i", 8), C.c_double: ("f", 8)}
    fields = []
    for name, ctype in cls._fields_:
        count = 1
        if hasattr(ctype, "_length_"):
            count, ctype = ctype._length_, ctype._type_
        if ctype not in kinds:
            fields.append((name, "other", 0, 1))
            continue
        fields.append((name,) + kinds[ctype] + (count,))
    return fields


@pytest.mark.parametrize("c_name,rust_name", [("relp_options", "RelpOptions"), ("relp_result", "RelpResult"),
                                              ("relp_exact_result", "RelpExactResult"), ("relp_bi_options", "RelpBiOptions"),
                                              ("relp_exact_width_record", "RelpExactWidthRecord")])
def test_integration_md_repr_c_structs_match_the_header(c_name, rust_name):
    assert rust_struct(rust_name) == c_struct(c_name)


def test_ctypes_classes_match_the_header():
    from relp_amd import api
    pairs = [("relp_options", api.Options), ("relp_result", api.Result), ("relp_exact_result", api.ExactResult), ("relp_stats", api.Stats),
             ("relp_batch_worker", api.BatchWorker), ("relp_exact_width_record", api.ExactWidthRecord)]
    for c_name, cls in pairs:
        assert ctypes_struct(cls) == c_struct(c_name), c_name
    # relp_batch_entry embeds a relp_result: names in order, and the sizes add up
    entry = [f[0].split()[-1] for f in c_struct("relp_batch_entry")]
    assert entry == [name for name, _ in api.BatchEntry._fields_]
    assert C.sizeof(api.BatchEntry) == 4 * 4 + C.sizeof(api.Result) + 2 * 8


def test_struct_sizes_have_no_hidden_padding_surprises():
    """Natural alignment of the declared fields, computed independently, equals ctypes' size (so Rust's repr(C) agrees as well)."""
    from relp_amd import api
    for c_name, cls in [("relp_options", api.Options), ("relp_result", api.Result), ("relp_exact_result", api.ExactResult)]:
        offset, largest = 0, 1
        for _, _, width, count in c_struct(c_name):
            offset = (offset + width - 1) // width * width
            offset += width * count
            largest = max(largest, width)
        offset = (offset + largest - 1) // largest * largest
        assert offset == C.sizeof(cls), c_name


def test_options_struct_size_is_checked_and_an_explicit_option_beats_the_environment(monkeypatch):
    """relp_options_default writes struct_size = sizeof(relp_options); relp_create refuses a struct that was not initialised by it
    (advisor, round 4: an older, shorter struct used to be over-read).  The binding's mapping of the old environment hooks never
    overrides a field the caller set -- and the library itself reads no such variable (grep: one getenv, the diagnostics switch)."""
    from relp_amd import api
    options = api.default_options()
    assert options.struct_size == C.sizeof(api.Options)
    handle = C.c_void_p()
    stale = api.default_options()
    stale.struct_size = 0
    assert api.lib().relp_create(C.byref(stale), C.byref(handle)) == api.ERR_ARGUMENT
    newer = api.default_options()
    newer.struct_size = C.sizeof(api.Options) + 8
    assert api.lib().relp_create(C.byref(newer), C.byref(handle)) == api.ERR_ARGUMENT
    monkeypatch.setenv("RELP_FTRAN_MIN_NNZ", "77")
    monkeypatch.setenv("RELP_NO_FUSED", "1")
    monkeypatch.setenv("RELP_NO_DENSE_LANE", "1")
    from_environment = api.default_options()
    assert (from_environment.ftran_min_nnz, from_environment.pivot_kernels) == (77, 1) and from_environment.switches & api.SW_NO_DENSE_LANE
    explicit = api.default_options(ftran_min_nnz=5, pivot_kernels=0, switches=0)
    assert (explicit.ftran_min_nnz, explicit.pivot_kernels, explicit.switches) == (5, 0, 0)
    sources = os.path.join(ROOT, "relp_amd", "csrc")
    calls = sum(open(os.path.join(sources, name)).read().count("getenv(") for name in os.listdir(sources) if name.endswith((".hip", ".cpp", ".hpp")))
    assert calls <= 2, calls  # model.hpp's `diagnostic` (stderr timelines only)


def test_a_caller_built_against_an_older_header_gets_no_byte_past_its_struct():
    """Advisor, round 5 (medium): relp_options_default used to memset the LIBRARY's sizeof(relp_options) into the caller's struct and to
    store the library's size.  Now the caller states its size (the header's macro does it for C callers): only that many bytes are
    written, struct_size is the caller's, relp_create reads that many, and a size no header ever had is refused."""
    from relp_amd import api
    lib = api.lib()
    full = C.sizeof(api.Options)
    round4 = api.Options.lu_refactor.offset + 4            # the struct as round 4's header had it
    buffer = (C.c_ubyte * (full + 64))(*([0xAB] * (full + 64)))
    options = C.cast(buffer, C.POINTER(api.Options))
    assert lib.relp_options_default_sized(options, round4) == api.OK
    assert options.contents.struct_size == round4 and options.contents.polish_period == 256 and options.contents.tol_dual == 1e-9
    assert all(b == 0xAB for b in bytes(buffer)[round4:]), "bytes past the caller's struct were written"
    # ... the full struct
    for k in range(len(buffer)):
        buffer[k] = 0xCD
    assert lib.relp_options_default_sized(options, full) == api.OK
    assert options.contents.struct_size == full and all(b == 0xCD for b in bytes(buffer)[full:])
    # sizes no header ever had: between the rounds (a field would be copied in half), zero, larger than the library's
    for size in (round4 + 4, round4 - 4, full - 4, 0, -8, full + 8):
        assert lib.relp_options_default_sized(options, size) == api.ERR_ARGUMENT, size
        stale = api.default_options()
        stale.struct_size = size
        handle = C.c_void_p()
        assert lib.relp_create(C.byref(stale), C.byref(handle)) == api.ERR_ARGUMENT, size
    assert lib.relp_options_default_sized(None, full) == api.ERR_ARGUMENT
    # the tuning fields are clamped to what the header promises (64 slices, a CU's 160 KB of LDS): checked where a handle can be made

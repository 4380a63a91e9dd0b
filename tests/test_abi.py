"""The C-ABI library loads and exports every symbol ``include/relp_amd.h`` declares (no compute calls; CPU only)."""
import os
import re

import pytest

import relp_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "relp_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(relp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = relp_amd.lib()
    names = declared_symbols()
    assert len(names) >= 40
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(relp_amd.SYMBOLS) == names


def test_version_string():
    assert relp_amd.lib().relp_version().decode().startswith("relp_amd")


def test_default_options_reproduce_the_reference_configuration():
    o = relp_amd.default_options()
    assert o.pivot_rule == relp_amd.STEEPEST_EDGE  # two_phase/mod.rs:57,68,107
    assert o.tol_dual > 0 and o.tol_pivot > 0


def test_no_cpu_fallback():
    """Without a HIP device the product must fail loudly instead of computing on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(relp_amd.RelpError) as info:
        relp_amd.Solver()
    assert info.value.status == relp_amd.api.ERR_DEVICE


def test_argument_errors():
    lib = relp_amd.lib()
    assert lib.relp_options_default(None) == relp_amd.api.ERR_ARGUMENT
    assert lib.relp_create(None, None) == relp_amd.api.ERR_ARGUMENT
    with pytest.raises(relp_amd.RelpError):
        relp_amd.Model(os.path.join(ROOT, "data", "does_not_exist.mps"))


def test_basis_inverse_object_has_no_cpu_fallback():
    """`relp_bi_identity` / `relp_bi_invert` need a HIP device too; only `relp_lu_factor_host` is host-only."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from relp_amd.basis_inverse import BasisInverse
    with pytest.raises(relp_amd.RelpError) as info:
        BasisInverse.identity(3)
    assert info.value.status == relp_amd.api.ERR_DEVICE
    with pytest.raises(relp_amd.RelpError) as info:
        BasisInverse.invert([[(0, 1.0)], [(1, 1.0)]])
    assert info.value.status == relp_amd.api.ERR_DEVICE


def test_batch_needs_a_device_too():
    """`relp_batch_create` builds handles with `relp_create`: no device, no batch (and nothing leaks)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    model = relp_amd.Model(os.path.join(ROOT, "data", "netlib", "AFIRO.SIF"))
    with pytest.raises(relp_amd.RelpError) as info:
        relp_amd.Batch([model])
    assert info.value.status == relp_amd.api.ERR_DEVICE

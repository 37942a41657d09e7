"""The C-ABI library loads and exports every symbol include/mia_letkf.h declares; the host-only
entry points (version, status strings, workspace queries, argument validation that returns before
any HIP call) behave as documented.  No GPU needed."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    import torch_assimilate_amd as mia
    mia.build()
    from torch_assimilate_amd import _cabi
    return _cabi.lib()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mia_letkf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mia_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from torch_assimilate_amd import _cabi
    decl = declared_symbols()
    assert len(decl) >= 20
    for name in decl:
        assert hasattr(lib, name), name
    assert sorted(_cabi.EXPORTED_SYMBOLS) == decl


def test_version_and_status_strings(lib):
    assert lib.mia_version() == 100
    assert lib.mia_status_string(0) == b"ok"
    for code in (-1, -2, -3, -4, -5, 700):
        assert len(lib.mia_status_string(code)) > 3


def test_workspace_queries(lib):
    n = C.c_size_t(0)
    assert lib.mia_letkf_localize_workspace_bytes(50000, 1, C.byref(n)) == 0 and n.value > 4 * 50000
    assert lib.mia_letkf_localize_workspace_bytes(10, 4, C.byref(n)) == -2        # too many coordinates
    assert lib.mia_letkf_localize_workspace_bytes(10, 1, None) == -1
    assert lib.mia_letkf_analysis_workspace_bytes(40, 50000, 4, C.byref(n)) == 0
    assert n.value >= 50000 * 44 * 4                                              # records of round_up(k+1, 4) floats
    assert lib.mia_letkf_analysis_workspace_bytes(40, 50000, 2, C.byref(n)) == -2
    assert lib.mia_etkf_workspace_bytes(20, 40, 8, C.byref(n)) == 0 and n.value >= (20 * 20 + 20) * 8
    assert lib.mia_etkf_workspace_bytes(1, 40, 8, C.byref(n)) == -2


def test_argument_validation_precedes_any_device_work(lib):
    # invalid sizes / NULL pointers are rejected on the host (negative codes), never reaching HIP
    assert lib.mia_gaspari_cohn_f64(None, -1, None, None) == -2
    assert lib.mia_gaspari_cohn_f64(None, 0, None, None) == 0
    assert lib.mia_gaspari_cohn_f32(None, 5, None, None) == -1
    assert lib.mia_gaspari_cohn_inf_f64(None, -1, None, None) == -2
    assert lib.mia_gaspari_cohn_inf_f32(None, 5, None, None) == -1
    # kernel expressions are validated on the host: NULL program, unknown opcode, operator without operands,
    # two values left on the stack, deeper than the device's operand stack
    from torch_assimilate_amd._cabi import KernelOp
    def prog(*ops):
        arr = (KernelOp * len(ops))()
        for i, (op, val) in enumerate(ops):
            arr[i].op, arr[i].value = op, val
        return arr
    def call(p, n):
        return lib.mia_lketkf_kernel_analysis_packed_f64(None, 10, 1, 4, 0, 5, None, 0, None, None, None, 8, 8, 1.0,
                                                         p, n, None, 10, 0, None, None, None)
    assert call(None, 1) == -1
    assert call(prog((99, 0.0)), 1) == -2
    assert call(prog((6, 0.0)), 1) == -2
    assert call(prog((1, 0.0), (2, 0.0)), 2) == -2
    assert call(prog(*[(4, 1.0)] * 7 + [(6, 0.0)] * 6), 13) == -3
    assert call(prog((1, 0.0), (4, 1.0), (6, 0.0)), 3) == -1          # well formed: fails later, on the NULL state
    assert lib.mia_letkf_localize_from_dist_taper_f64(7, None, None, 0, 8, None, 1, 1e-5, None, None, None, None, None) == -2
    assert lib.mia_apply_weights_f32(None, 10, 1, 1, 0, 5, None, None, 10, 0, None) == -2     # k < 2
    assert lib.mia_apply_weights_f32(None, 10, 1, 4, 0, 0, None, None, 10, 0, None) == 0      # empty shard
    assert lib.mia_apply_weights_f32(None, 10, 1, 4, 0, 5, None, None, 10, 0, None) == -1
    assert lib.mia_letkf_pack_obs_f32(None, None, 4, 0, None, None) == 0
    assert lib.mia_letkf_pack_obs_f32(None, None, 4, 8, None, None) == -1
    assert lib.mia_letkf_analysis_packed_f32(None, 10, 1, 4, 0, 0, None, 0, None, None, None, 8, 4, 1.0, 0.0,
                                             None, 10, 0, None, None, None) == 0
    assert lib.mia_letkf_analysis_packed_f32(None, 10, 1, 4, 0, 5, None, 0, None, None, None, 8, 4, -1.0, 0.0,
                                             None, 10, 0, None, None, None) == -2             # inf_factor <= 0


def test_engine_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import torch_assimilate_amd as mia
    with pytest.raises(mia.MiaError):
        mia.LetkfEngine()


def test_route_options(lib):
    """mia_set_option / mia_get_option: names, defaults, ranges (no device work)."""
    v = C.c_int(-5)
    for name, default in ((b"cheb_dmax", 62), (b"cheb_table", 1), (b"cheb_rowbatch", 1), (b"cheb_big", 1), (b"tile", 1), (b"tile_split", 1), (b"localize_quad", 1), (b"step_hostwait", 1), (b"step_lazy_sort", 1),
                          (b"segment_signal", 1)):
        assert lib.mia_get_option(name, C.byref(v)) == 0 and v.value == default
    assert lib.mia_set_option(b"cheb_dmax", 14) == 0 and lib.mia_get_option(b"cheb_dmax", C.byref(v)) == 0 and v.value == 14
    assert lib.mia_set_option(b"cheb_dmax", 2) == -2 and lib.mia_set_option(b"cheb_dmax", 63) == -2
    assert lib.mia_set_option(b"cheb_dmax", -1) == 0 and lib.mia_get_option(b"cheb_dmax", C.byref(v)) == 0 and v.value == 62
    assert lib.mia_set_option(b"tile", 7) == 0 and lib.mia_get_option(b"tile", C.byref(v)) == 0 and v.value == 1
    assert lib.mia_set_option(b"tile", 0) == 0 and lib.mia_get_option(b"tile", C.byref(v)) == 0 and v.value == 0
    assert lib.mia_set_option(b"tile", -1) == 0 and lib.mia_get_option(b"tile", C.byref(v)) == 0 and v.value == 1
    assert lib.mia_set_option(b"no_such_option", 1) == -3 and lib.mia_get_option(b"no_such_option", C.byref(v)) == -3
    assert lib.mia_set_option(None, 1) == -1 and lib.mia_get_option(b"tile", None) == -1

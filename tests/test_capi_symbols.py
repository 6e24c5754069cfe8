"""CPU: the C-ABI library builds for gfx950, loads, and exports exactly what include/dvmvs_hip.h declares.
No compute call is made (there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
HEADER = os.path.join(ROOT, "include", "dvmvs_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dvmvs_[a-z0-9_]+)\s*\(", text)))


def declared_parameter_counts():
    """{symbol: number of parameters} from the header's prototypes (``(void)`` = 0)."""
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    counts = {}
    for name, params in re.findall(r"\b(dvmvs_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        params = params.strip()
        counts[name] = 0 if params in ("", "void") else params.count(",") + 1
    return counts


@pytest.fixture(scope="module")
def library():
    from dvmvs.hip import _capi
    if not os.path.exists(_capi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _capi


def test_header_declares_the_four_ops():
    syms = declared_symbols()
    for op in ("cost_volume", "hidden_warp", "lstm_gates"):
        assert f"dvmvs_{op}_fwd" in syms and f"dvmvs_{op}_bwd" in syms
    assert "dvmvs_depth_reproject_fwd" in syms and "dvmvs_relative_pose" in syms and "dvmvs_sweep_matrices" in syms


def test_library_exports_every_declared_symbol(library):
    handle = ctypes.CDLL(library.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(handle, name), f"{name} declared in dvmvs_hip.h but not exported by libdvmvs_hip.so"


def test_binding_table_matches_header(library):
    assert sorted(library.SIGNATURES) == declared_symbols()
    counts = declared_parameter_counts()
    for name, (_, argtypes) in library.SIGNATURES.items():
        assert len(argtypes) == counts[name], f"{name}: {len(argtypes)} ctypes arguments for {counts[name]} declared parameters"
    lib = library.lib()
    assert lib.dvmvs_abi_version() == library.ABI_VERSION
    assert lib.dvmvs_build_arch() == b"gfx950"
    assert b"invalid argument" in lib.dvmvs_error_string(-1)


def test_argument_validation_without_gpu(library):
    """Negative return codes are produced before anything is enqueued, so this is safe without a device."""
    lib = library.lib()
    null = None
    assert lib.dvmvs_relative_pose(null, null, null, 1, null) == -1
    assert lib.dvmvs_lstm_gates_fwd(null, null, null, null, 1, 512, 8, 10, null) == -1
    assert lib.dvmvs_hidden_warp_fwd(null, null, null, null, null, 1, 512, 8, 10, 1, null) == -1
    arr = library.pointer_array([None])
    assert lib.dvmvs_cost_volume_fwd(null, arr, null, null, null, 1, 1, 32, 128, 160, 64, 0.25, 20.0, 1, 0, 0, null, 0, null) == -1
    assert lib.dvmvs_sweep_matrices(null, arr, null, null, null, 1, 1, null) == -1
    assert lib.dvmvs_depth_reproject_fwd(null, null, null, null, null, null, 16, 1, 256, 320, null) == -1
    assert lib.dvmvs_cost_volume_workspace_bytes(0, 3, 128, 160, 64) == 0


def test_workspace_sizes(library):
    lib = library.lib()
    # spill workspace of the two-pass sweep: 4 header words + per possible work item one id and one slot of (1 + 8 M) words; a launch
    # has at most twice as many work items as (32x8-pixel tile, 8-plane chunk) pairs (dvmvs_sweep_work_list cuts long ones)
    groups = 2 * (160 // 32) * (128 // 8) * 8
    assert lib.dvmvs_sweep_work_list_bytes(1, 128, 160, 64) == 4 * (2 + 2 * groups)
    assert lib.dvmvs_cost_volume_workspace_bytes(1, 2, 128, 160, 64) == 4 * (4 + groups + groups * (1 + 2 * 8))
    assert lib.dvmvs_cost_volume_workspace_bytes(0, 2, 128, 160, 64) == 0
    assert lib.dvmvs_cost_volume_workspace_bytes(2, 3, 33, 47, 10) > 0


def test_code_object_is_gfx950(library):
    blob = open(library.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"gfx942" not in blob and b"sm_" not in blob

"""CPU-side checks of the drop-in boundary: the library loads and exports exactly what
include/daliti_s2m.h declares, and fails loudly (no CPU fallback) without a gfx950 device."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import daliti_amd
    if not os.path.exists(daliti_amd.library_path()):
        daliti_amd.build_library()
    return daliti_amd.load_library()


def _declared():
    hdr = open(os.path.join(ROOT, "include", "daliti_s2m.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(s2m_[a-z0-9_]+)\s*\(", hdr)))


def test_every_declared_symbol_is_exported(lib):
    from daliti_amd import ABI_SYMBOLS
    decl = _declared()
    assert len(decl) >= 25
    assert sorted(ABI_SYMBOLS) == decl          # the Python binding tracks the header
    for name in decl:
        assert hasattr(lib, name), name


def test_abi_version_and_defaults(lib):
    from daliti_amd.engine import Config
    assert lib.s2m_abi_version() == 5
    cfg = Config()
    assert lib.s2m_config_default(C.byref(cfg)) == 0
    # reference constants (laserMapping.cpp:76-77, 853, 863, 870, 889, 1040; feat.yaml)
    assert abs(cfg.plane_thr - 0.1) < 1e-7 and cfg.knn_d2_gate == 5.0
    assert cfg.s_gate == 0.9 and cfg.res_gate == 2.0 and cfg.laser_point_cov == 0.0015
    assert cfg.conv_rot_deg == 0.01 and cfg.conv_pos_cm == 0.015
    assert cfg.max_iter == 10 and cfg.feat_threshold == 100 and cfg.extrinsic_est_en == 0
    assert lib.s2m_config_default(None) != 0
    assert lib.s2m_strerror(-2) == b"no gfx950 HIP device"


def test_struct_layouts_match_header(lib):
    from daliti_amd.engine import Config, PassOut, IterLog, DynShare
    assert C.sizeof(Config) == 96
    assert C.sizeof(PassOut) == 144 * 8 + 12 * 8 + 8 + 8
    assert C.sizeof(IterLog) == 16 + 3 * 64 * 4 + 64 * 8 + 64 * 24 * 8
    assert C.sizeof(DynShare) == 48


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from daliti_amd import Engine, S2MError
    with pytest.raises(S2MError) as ei:
        Engine()
    assert ei.value.code == -2                  # S2M_ERR_NO_DEVICE, not a silent CPU path


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "daliti_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "s2m_oracle" not in src.replace("oracle/s2m_oracle.c", ""), f


def test_mirror_header_compiles_against_the_abi(tmp_path):
    """include/daliti_s2m_mirror.hpp (the node's copy of the map, fed by s2m_map_get_changes) is header-only C++ over the C ABI:
    it compiles on its own with plain g++ and uses nothing the header does not declare."""
    import subprocess
    src = tmp_path / "mirror_tu.cpp"
    src.write_text('#include "daliti_s2m_mirror.hpp"\nint use(s2m_engine *e) { static s2m_map_mirror m; return m.update(e); }\n')
    subprocess.check_call(["g++", "-std=c++14", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)])
    hpp = open(os.path.join(ROOT, "include", "daliti_s2m_mirror.hpp")).read()
    used = set(re.findall(r"\b(s2m_[a-z0-9_]+)\s*\(", hpp))
    assert {"s2m_map_get_changes", "s2m_map_get_points", "s2m_map_get_ids"} <= used and used <= set(_declared())

"""CPU-only: the C-ABI library is built, loads, and exports every symbol include/lmc_atomi.h
declares (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lmc_atomi.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lmc_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_surface():
    fns = declared_functions()
    for must in ("lmc_version", "lmc_last_error", "lmc_myula_create", "lmc_sampler_step", "lmc_sampler_destroy",
                 "lmc_fused_eval", "lmc_blur", "lmc_gradient", "lmc_gradient_adjoint", "lmc_energies",
                 "lmc_prox_elementwise", "lmc_dual_project", "lmc_sampler_get_moments"):
        assert must in fns


def test_library_loads_and_exports_every_declared_symbol():
    from lmc_atomi_amd import _capi
    lib = _capi.load()
    assert lib.lmc_version() == _capi.ABI_VERSION
    raw = ctypes.CDLL(_capi.LIB_PATH)
    for name in declared_functions():
        assert hasattr(raw, name), f"{name} declared in lmc_atomi.h but not exported"
    # the ctypes binding covers the whole header, nothing more
    assert sorted(_capi.exported_symbols()) == declared_functions()


def test_struct_layouts_match_the_header_sizes():
    # sizes computed by the C compiler for the same declarations
    import subprocess
    import tempfile
    from lmc_atomi_amd import _capi
    code = '#include <stdio.h>\n#include "lmc_atomi.h"\nint main(){printf("%zu %zu %zu\\n", sizeof(lmc_problem), sizeof(lmc_myula_config), sizeof(lmc_ulpda_config));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "s.c"), os.path.join(d, "s")
        open(src, "w").write(code)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe], check=True)
        a, b, c = map(int, subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split())
    assert ctypes.sizeof(_capi.lmc_problem) == a
    assert ctypes.sizeof(_capi.lmc_myula_config) == b
    assert ctypes.sizeof(_capi.lmc_ulpda_config) == c


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np
    import lmc_atomi_amd as la
    H = la.Convolve2D((8, 8), np.ones((5, 5)) / 25)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        H.matvec(np.zeros(64))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        la.MoreauYosidaUnadjustedLangevin(la.L2(Op=H, b=np.zeros(64), sigma=1.0), la.TV((8, 8), 0.3), np.zeros(64),
                                          tau=0.1, gamma=0.5, niter=2)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "lmc_atomi_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("no oracle", ""), f"{f} mentions the oracle"

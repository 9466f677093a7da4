"""CPU-only fence around the scratch (private segment) sizes of the SHIPPED library.

Round 2 saw one memory access fault on the GPU box (DESIGN 3.0p "Scratch finding"): a kernel with 292 B of scratch per lane followed by one
with 268 B.  Whatever the runtime does there, the library's answer is structural -- no kernel may need more than SCRATCH_LIMIT bytes per
lane, and the kernels the reference's own models reach at 512 x 512 may need none -- and this test is what enforces it: it reads
`private_segment_fixed_size` of every kernel from the code-object notes of liblmc_atomi.so (scripts/kernel_resources.py)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))

SCRATCH_LIMIT = 256     # bytes per lane; 256 B x 64 lanes x 8192 wave slots = 128 MiB, below the runtime's single-allocation scratch limit

# kernels a model of the reference's driver (prox_lmc_deconv.py:447-703: 5x5 / 6x6 / 7x7 box blurs, TV niter = 10, MC-TV, ME-TV with
# niter_l2 = 50) reaches at 512 x 512 -- MYULA and ULPDA -- plus the BASELINE configurations (l2 prior, mask + Haar): no scratch at all
# (template arguments: pipe <K, PXL, KT, CHAIN, WARM, AL, RT>; rows <PXL, KT, DOT, ULO, UHI, AL, EP>)
ZERO_SCRATCH = [
    "myula_step_pipe_kernel<10, 8, 5, false, false, true, false>",     # 5x5 blur + TV / MC-TV / ME-TV step (headline)
    "myula_step_pipe_kernel<10, 8, 7, false, false, true, false>",     # 6x6 / 7x7 blur
    "myula_step_pipe_kernel<10, 8, 5, false, false, true, true>",      # the same with the per-chain early exit (tv_rtol = 1e-4, as the reference is configured)
    "myula_step_pipe_kernel<10, 8, 7, false, false, true, true>",
    "myula_step_pipe_kernel<10, 8, 0, true, false, true, false>",      # links of the ME-TV inner prox
    "myula_step_pipe_kernel<10, 8, 0, true, false, true, true>",
    "myula_step_rows_kernel<8, 5, true, -1, -1, true, false>",                # ULPDA: operator of the implicit step (first Chebyshev iteration, CG)
    "myula_step_rows_kernel<8, 7, true, -1, -1, true, false>",
    "myula_step_rows_kernel<8, 5, false, 0, 4, true, false>",                 # uniform boxes: 5x5, 6x6 (window 0..5 of 7 taps), 7x7
    "myula_step_rows_kernel<8, 7, false, 0, 5, true, false>",
    "myula_step_rows_kernel<8, 7, false, 0, 6, true, false>",
    "myula_step_rows_pair_kernel<8>",
    "cheb_pair_kernel<8",
    "myula_step_block_kernel<3, 5, false",                             # mask + Haar
    "moments4_kernel",
]


@pytest.fixture(scope="module")
def resources():
    import kernel_resources
    from lmc_atomi_amd import _capi
    return kernel_resources.kernel_resources(_capi.LIB_PATH)


def test_every_kernel_is_listed(resources):
    names = [r["demangled"] for r in resources]
    assert len(names) > 100
    for must in ("myula_step_pipe_kernel<10, 8, 5, false, false, true, false>", "moments4_kernel", "cheb_pair_kernel<8, false>"):
        assert any(n.startswith(must) for n in names), must


def test_no_kernel_exceeds_the_scratch_limit(resources):
    worst = sorted(resources, key=lambda r: -r["scratch"])[:5]
    over = [(r["demangled"], r["scratch"]) for r in resources if r["scratch"] > SCRATCH_LIMIT]
    assert not over, f"kernels above {SCRATCH_LIMIT} B of scratch per lane: {over}; worst five: {[(r['demangled'], r['scratch']) for r in worst]}"


def test_kernels_of_the_reference_models_use_no_scratch(resources):
    bad = []
    for prefix in ZERO_SCRATCH:
        hits = [r for r in resources if r["demangled"].startswith(prefix)]
        assert hits, f"no kernel matches {prefix}"
        bad += [(r["demangled"], r["scratch"]) for r in hits if r["scratch"] > 0]
    assert not bad, f"kernels on the reference's model paths that spill to scratch: {bad}"


def test_role_pairing_codes_are_permutations():
    """The pipe kernel assigns its eight roles to hardware waves by 8-nibble codes (nibble w = role of hardware wave w; waves w and w + 4 share a
    SIMD): every code in the kernel source must name each role exactly once -- a repeated role would leave another one without a wave."""
    import re
    src = open(os.path.join(ROOT, "lmc_atomi_amd", "csrc", "lmc_step_pipe_kernel.h")).read()
    codes = re.findall(r"0x([0-9a-fA-F]{8})u", src)
    assert len(codes) >= 5
    for c in codes:
        assert sorted(int(ch, 16) for ch in c) == list(range(8)), c

"""Build-time guard on the hot kernels' register budget (no GPU needed: hipcc cross-compiles gfx950 here).

Occupancy is part of the design: k_wf_ext is launched 6 blocks of 256 threads per CU, which needs
<= 80 VGPRs per lane, and neither it nor k_wf_shade may spill.  A harmless-looking edit can break that silently -- pinning
NaN -> texel 0 as `x == x ? (int)x : 0` made k_wf_shade spill 592 bytes per lane and cost 36 ms per frame (21 -> 58 ms); the
same thing written as fmaxf(x, 0) costs nothing.  This test reads hipcc's own resource report."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def usage(tmp_path_factory):
    out = tmp_path_factory.mktemp("res") / "x.o"
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value",
           "-I" + os.path.join(ROOT, "include"), "-c", os.path.join(ROOT, "hobbyraytracer_amd", "csrc", "hrt_hip.hip"), "-o", str(out),
           "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    res, name = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1); res[name] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m and name:
            res[name][m.group(1).split(" ")[0]] = int(m.group(2))
    assert res, "hipcc printed no resource report"
    return res


def _find(usage, fragment):
    hits = {k: v for k, v in usage.items() if fragment in k}
    assert hits, fragment
    return hits


def test_traversal_kernel_keeps_six_waves_per_simd(usage):
    # k_wf_ext<false, 20 | 24>: the variants every mesh up to BVH depth 24 runs (the headline teapot: depth 18)
    for frag in ("8k_wf_extILb0ELi20EE", "8k_wf_extILb0ELi24EE"):
        for name, u in _find(usage, frag).items():
            assert u["ScratchSize"] == 0, (name, u)
            assert u["VGPRs"] <= 80, (name, u)       # 512 / 80 -> 6 waves per SIMD = the 6 blocks per CU that are launched
            assert u["LDS"] <= 160 * 1024 // 6, (name, u)


def test_no_hot_kernel_spills(usage):
    for frag in ("8k_wf_extILb0", "10k_wf_shadeILb0", "8k_wf_genILb0", "8k_wf_preILb0", "9k_wf_tailILb0ELi32", "11k_pathtraceILb0", "10k_wf_stale"):
        for name, u in _find(usage, frag).items():
            assert u["ScratchSize"] == 0, (name, u)
    for name, u in _find(usage, "10k_wf_shadeILb0").items():
        assert u["VGPRs"] <= 128, (name, u)          # 4 waves per SIMD


def test_tail_kernel_keeps_four_blocks_per_cu(usage):
    # k_wf_tail<false, 20 | 24>: residency is what it lives on (one wave per task of a small batch: DESIGN.md 5) -- four blocks of four
    # waves per CU need <= 128 VGPRs and <= 40 KB of LDS; the register cap costs 12 bytes of scratch, bought knowingly (10.9 -> 10.0 ms on
    # the 1/8 share of the headline frame); more would say the kernel grew
    for frag in ("9k_wf_tailILb0ELi20EE", "9k_wf_tailILb0ELi24EE"):
        for name, u in _find(usage, frag).items():
            assert u["VGPRs"] <= 128 and u["ScratchSize"] <= 16, (name, u)
            assert 4 * u["LDS"] <= 160 * 1024, (name, u)


def test_instrumented_builds_still_compile():
    """The diagnostic builds (-DHRT_DEBUG_BOUNDS: tests/tools/debug_bounds.sh; -DHRT_TA_PROBE / -DHRT_VALU_PROBE: tests/tools/ta_probe.sh)
    are not part of the product build: a syntax-only pass keeps them from rotting."""
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value",
           "-DHRT_DEBUG_BOUNDS", "-DHRT_TA_PROBE=1", "-DHRT_VALU_PROBE=2", "-fsyntax-only",
           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "hobbyraytracer_amd", "csrc", "hrt_hip.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]

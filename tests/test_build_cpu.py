"""The HIP sources cross-compile for gfx950 on a host without a GPU, and the hot kernels keep the register
budget the design depends on: no scratch spills, the SSIM gradient kernel at <= 168 VGPRs (3 waves per
SIMD), everything else at <= 128 (4 waves per SIMD).  A spill turns the 85 us kernel into a 235 us one."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sfm-learner-chainer_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


LOSS_UNITS = ["sfm_loss.hip", "sfm_loss_ref.hip", "sfm_loss_dsrc.hip", "sfm_loss_pair.hip"]      # the translation units that instantiate loss_body


def _resource_usage(tmp_path):
    """kernel name -> {remark: value} over the three translation units, compiled side by side (device code only)."""
    flags = None
    for line in open(os.path.join(CSRC, "Makefile")):
        if line.startswith("CXXFLAGS"):
            flags = line.split("=", 1)[1].replace("$(ARCH)", "gfx950").split()
    assert flags and "-fno-slp-vectorize" in flags
    procs = []
    for unit in LOSS_UNITS:
        cmd = [HIPCC] + [f for f in flags if f != "-fPIC"] + ["-c", unit, "-o", str(tmp_path / (unit + ".o")),
                                                             "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only"]
        procs.append((unit, subprocess.Popen(cmd, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    kernels = {}
    for unit, pr in procs:
        _, err = pr.communicate(timeout=580)
        assert pr.returncode == 0, (unit, err[-2000:])
        name = None
        for line in err.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = m.group(1)
                kernels[name] = {"unit": unit}
                continue
            m = re.search(r"remark:\s+([\w /\[\]]+?):\s+(\d+)", line)
            if m and name:
                kernels[name][m.group(1).strip()] = int(m.group(2))
    return kernels


@pytest.mark.timeout(600)
def test_loss_kernels_fit_their_register_budget(tmp_path):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    kernels = _resource_usage(tmp_path)
    loss = {k: v for k, v in kernels.items() if "loss_kernel" in k}
    # sfm_loss.hip: {fwd, bwd, fused} x {l1, ssim, explain} x {no, 2nd-order, edge-aware smoothness} x {planar, hwc}, and the L1
    # gradient kernels ({bwd, fused} x 3 x 2) a second time for three waves per SIMD (loss_kernel_wide: small launches); the {fwd,
    # fused} ones of both kinds once more with the warped-image output (SfmLossDesc.warped, ABI v4): 54 + 12 + 36 + 6;
    # sfm_loss_dsrc.hip: the gradient kernels of a launch that also produces dL/d(src) ({bwd, fused, fused + warped} x 3 x 3 x 2 = 54:
    # the kernels of sfm_loss.hip plus three stores per pixel row, the record of dL/dI^) and the second launch, dsrc_scatter_kernel;
    # sfm_loss_ref.hip (round 6, ABI v5): every launch of the first kind in the reference's evaluation order (54 + 36 = 90)
    by_unit = lambda u: sum(1 for v in loss.values() if v["unit"] == u)
    assert by_unit("sfm_loss.hip") == 54 + 12 + 36 + 6
    assert by_unit("sfm_loss_dsrc.hip") == 54
    assert by_unit("sfm_loss_ref.hip") == 54 + 36
    # sfm_loss_pair.hip (round 6): the SSIM gradient kernels of the pixel-interleaved layout that walk two sources per pass
    # ({bwd, fused} x 3 smoothness forms), two waves per SIMD
    assert by_unit("sfm_loss_pair.hip") == 6
    for k, v in kernels.items():
        assert v["VGPRs Spill"] == 0, (k, v)
        # no frame at all, in any variant: scalar registers that do not fit are parked in vector-register lanes (counted in the
        # budget below), never in memory -- a non-zero frame means real scratch traffic or an array demoted to memory
        assert v["ScratchSize [bytes/lane]"] == 0, (k, v)
    for k, v in loss.items():
        if "loss_kernel_pair" in k:                                          # two sources per pass: two waves per SIMD
            budget = 256
        elif "loss_kernel_dsrc" in k:                                        # <SSIM, ...>: the occupancy of the kernels without d_src
            budget = 168 if "loss_kernel_dsrcILb1E" in k else 128
        else:
            ssim_grad = "loss_kernelILb1ELb1E" in k or "loss_kernel_wide" in k or "loss_kernel_refILb1ELb1E" in k   # three waves per SIMD
            budget = 168 if ssim_grad else 128
        assert v["VGPRs"] <= budget, (k, v["VGPRs"])
    # the second launch of a call with d_src: 12 wavefronts per workgroup, one workgroup per CU (its LDS window): no register limit in sight
    # (one per projection: it re-projects the pixels by the chain of the main launch)
    scat = [v for k, v in kernels.items() if "dsrc_scatter_kernel" in k]
    assert len(scat) == 2 and all(v["unit"] == "sfm_loss_dsrc.hip" and v["VGPRs"] <= 128 for v in scat), scat
    # the benchmarked kernel itself: its allocation is what the occupancy of DESIGN.md 4.1 rests on
    head = [v for k, v in loss.items() if "loss_kernelILb1ELb1ELb1ELb0ELi2ELb1ELb0E" in k]
    assert len(head) == 1 and head[0]["VGPRs"] <= 160 and head[0]["SGPRs Spill"] <= 5, head


@pytest.mark.timeout(600)
def test_committed_issue_model_is_the_one_of_these_sources():
    """bench.py's `roofline_valu` prices the counted vector instructions with profiles/r06_issue_model.json: the file must be the
    model of the kernels as they are in the tree (tools/issue_model.py --check recompiles and compares)."""
    import sys
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "issue_model.py"), "r06", "--check"], capture_output=True, text=True, timeout=580)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])

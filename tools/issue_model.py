#!/usr/bin/env python3
"""Vector-instruction issue model of the main kernel's row loop, from the compiled ISA and the measured issue costs.

    python tools/issue_model.py [tag]      (CPU only: hipcc -S; writes profiles/<tag>_issue_model.json and .md)

For every `loss_kernel<SSIM, GRAD, LOSS, EXPL, SMODE, HWC>` instantiation that bench.py can launch, the innermost loop with
the most instructions (the row loop of the source pass: three statically rotated row steps) is priced instruction by
instruction with the per-SIMD issue cost of its class at the kernel's occupancy, as measured by tools/op_cost.hip
(profiles/<tag>_op_cost_microbench.txt, column `k w: ... /SIMD`).  Rarely executed blocks inside that loop (the optional
d_src scatter with its global atomics, the out-of-image zero fill) are left out.  Result per kernel: vector instructions and
issue cycles of ONE row step, i.e. the mean issue cost of a vector instruction of this kernel's mix -- what bench.py multiplies
the counted SQ_INSTS_VALU of a launch with to get the issue-bound floor of that launch (`roofline_valu`).
"""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK = "--check" in sys.argv          # recompute from the sources and compare with the committed JSON instead of writing it
_args = [a for a in sys.argv[1:] if not a.startswith("--")]
tag = _args[0] if _args else "r06"
CSRC = os.path.join(ROOT, "sfm-learner-chainer_amd", "csrc")

# instruction -> row of the op-cost table whose measured cost it takes
CLASS_OF = [
    (r"^v_pk_", "v_pk_mul_f32"),
    (r"^v_(rcp|exp|log|rsq|sqrt|sin|cos)_", "v_rcp_f32"),
    (r"^v_cndmask", "v_cndmask_b32_e64 (sgpr mask)"),
    (r"^v_cmp", "v_cmp_lt_f32_e64 -> sgpr"),
    (r"^v_cvt|^v_floor|^v_trunc|^v_rndne", "v_floor_f32"),
    (r"^v_fract", "v_fract_f32"),
    (r"^v_(mul_lo|mul_hi|mad_u32|mad_i32|mul_u32)", "v_mul_lo_u32"),
    (r"^v_mad_u64|^v_mad_i64", "v_mad_u64_u32"),
    (r"^v_lshl_add_u64", "v_lshl_add_u64"),
    (r"^v_(lshl_add|add_lshl|lshl_or|and_or|add3|or3|xad)", "v_lshl_add_u32"),
    (r"^v_(lshlrev|lshrrev|ashrrev)", "v_lshlrev_b32"),
    (r"^v_(max|min)", "v_max_f32"),
    (r"^v_med3", "v_med3_f32"),
    (r"^v_bfi|^v_bfe|^v_perm", "v_bfi_b32"),
    (r"^v_readlane|^v_readfirstlane|^v_writelane", "v_cndmask_b32_e64 (sgpr mask)"),
    (r"^v_(add|sub)_u32|^v_(add|sub)_co", "v_add_u32"),
    (r"^v_mov_b64", "v_pk_mul_f32"),
    (r"^v_", "v_fma_f32"),
]


def op_costs(waves):
    path = os.path.join(ROOT, "profiles", "%s_op_cost_microbench.txt" % tag)
    costs = {}
    for line in open(path):
        m = re.match(r"^(.*?)\s+1w:.*?%dw:\s*[\d.]+ /wave\s+([\d.]+) /SIMD" % waves, line)
        if m:
            costs[m.group(1).strip()] = float(m.group(2))
    return costs, os.path.relpath(path, ROOT)


def price(ins, costs):
    op = ins.split()[0]
    if not op.startswith("v_"):
        if op.startswith("s_nop"):
            return "s_nop", costs["s_nop 0"]
        return None, 0.0
    if re.search(r"\b(row_|wave_|quad_perm)", ins):
        return "dpp", costs["v_add_f32_dpp wave_shr:1"]
    for pat, row in CLASS_OF:
        if re.match(pat, op):
            return row, costs[row]
    return "v_fma_f32", costs["v_fma_f32"]


def main():
    flags = None
    for line in open(os.path.join(CSRC, "Makefile")):
        if line.startswith("CXXFLAGS"):
            flags = line.split("=", 1)[1].replace("$(ARCH)", "gfx950").split()
    asm = "/tmp/sfm_loss_issue_model.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + [f for f in flags if f != "-fPIC"] + ["-S", "--cuda-device-only", "sfm_loss.hip", "-o", asm],
                          cwd=CSRC, stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN3sfm(11loss_kernel|16loss_kernel_wide)I.*:", l)]
    out = {"tag": tag, "kernels": {}}
    for n, st in enumerate(starts):
        name = lines[st].split(":")[0]
        # (the last template argument, WARPED, selects the variants that also write SfmLossDesc.warped: parity tests only, not priced)
        m = re.match(r"_ZN3sfm11loss_kernelILb(\d)ELb(\d)ELb(\d)ELb(\d)ELi(\d)ELb(\d)ELb(\d)EEE", name)
        wide = m is None
        if wide:      # loss_kernel_wide<LOSS, SMODE, HWC, WARPED>: the L1 gradient kernels built for three waves per SIMD
            mw = re.match(r"_ZN3sfm16loss_kernel_wideILb(\d)ELi(\d)ELb(\d)ELb(\d)EEE", name)
            ssim, grad, loss, expl, smode, hwc, warped = 0, 1, int(mw.group(1)), 0, int(mw.group(2)), int(mw.group(3)), int(mw.group(4))
        else:
            ssim, grad, loss, expl, smode, hwc, warped = [int(v) for v in m.groups()]
        if warped:
            continue
        body = lines[st:(starts[n + 1] if n + 1 < len(starts) else len(lines))]
        waves = 3 if ((ssim and grad) or wide) else 4
        costs, src = op_costs(waves)
        # basic blocks with their innermost-loop annotation
        blocks, cur = [], None
        for l in body:
            mb = re.match(r"^(\.LBB\S+):\s*(;.*)?$", l)
            if mb:
                hdr = re.search(r"Header=(\S+) Depth=(\d+)", l)
                own = re.search(r"Loop Header: Depth=(\d+)", l)
                cur = {"loop": (hdr.group(1), int(hdr.group(2))) if hdr else ((mb.group(1)[3:], int(own.group(1))) if own else None), "ins": []}
                blocks.append(cur)
                continue
            s = l.strip()
            if cur is None or not s or s[0] in ";.":
                continue
            cur["ins"].append(s.split(";")[0].strip())
        loops = collections.defaultdict(list)
        for b in blocks:
            if b["loop"] and b["loop"][1] >= 2:
                loops[b["loop"]].append(b)
        if not loops:
            continue
        key = max(loops, key=lambda k: sum(len(b["ins"]) for b in loops[k]))
        n_valu, cyc, classes = 0, 0.0, collections.Counter()
        for b in loops[key]:
            ins = b["ins"]
            if any(i.startswith("global_atomic") for i in ins):   # the optional d_src scatter: only what precedes its branch runs
                cut = [j for j, i in enumerate(ins) if i.startswith("s_cbranch") or i.startswith("s_and_saveexec")]
                ins = ins[:cut[0]] if cut else []
            if sum(1 for i in ins if i.startswith("v_mov_b32") and i.rstrip().endswith(", 0")) >= 12:
                continue                                   # zero fill of a ring slot outside the image (rare)
            for i in ins:
                c, p = price(i, costs)
                if c is None:
                    continue
                cyc += p
                classes[c] += 1
                if i.startswith("v_"):
                    n_valu += 1
        steps = 3.0 if ssim else 1.0                       # the SSIM pass instantiates the row step three times (ring rotation)
        full = "void sfm::loss_kernel<%s>(sfm::LossArgs)" % ", ".join(
            [("true" if v else "false") for v in (ssim, grad, loss, expl)] + [str(smode), "true" if hwc else "false", "false"])
        if wide:
            full = "void sfm::loss_kernel_wide<%s, %d, %s, false>(sfm::LossArgs)" % ("true" if loss else "false", smode, "true" if hwc else "false")
        out["kernels"][full] = {
            "waves_per_simd": waves, "valu_per_row_step": round(n_valu / steps, 1), "issue_cycles_per_row_step": round(cyc / steps, 1),
            "mean_issue_cycles_per_valu": round(cyc / max(n_valu, 1), 4),
            "classes_per_row_step": {k: round(v / steps, 1) for k, v in sorted(classes.items())},
            "op_cost_source": src}
    if CHECK:
        have = json.load(open(os.path.join(ROOT, "profiles", "%s_issue_model.json" % tag)))["kernels"]
        stale = [k for k, v in out["kernels"].items() if k not in have or have[k]["valu_per_row_step"] != v["valu_per_row_step"]
                 or have[k]["issue_cycles_per_row_step"] != v["issue_cycles_per_row_step"]]
        if stale:
            print("profiles/%s_issue_model.json is stale for %d kernels, e.g. %s: committed %s, sources %s" % (
                tag, len(stale), stale[0], have.get(stale[0]), out["kernels"][stale[0]]))
            sys.exit(1)
        print("profiles/%s_issue_model.json matches the sources (%d kernels)" % (tag, len(out["kernels"])))
        return
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "profiles", "%s_issue_model.json" % tag), "w"), indent=1, sort_keys=True)
    with open(os.path.join(ROOT, "profiles", "%s_issue_model.md" % tag), "w") as f:
        f.write("# Issue model of the row step (%s): ISA of HEAD priced with %s\n\n| kernel | waves/SIMD | VALU per row step | issue cycles per row step (per SIMD) | cycles per VALU |\n|---|---|---|---|---|\n" % (tag, src))
        for k, v in sorted(out["kernels"].items()):
            f.write("| `%s` | %d | %.0f | %.0f | %.3f |\n" % (k.replace("void sfm::", "").replace("(sfm::LossArgs)", ""), v["waves_per_simd"], v["valu_per_row_step"],
                                                         v["issue_cycles_per_row_step"], v["mean_issue_cycles_per_valu"]))
    k = "void sfm::loss_kernel<true, true, true, false, 2, true, false>(sfm::LossArgs)"
    print(k, json.dumps(out["kernels"].get(k), indent=1))


if __name__ == "__main__":
    main()

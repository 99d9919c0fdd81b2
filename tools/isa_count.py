#!/usr/bin/env python3
"""Instruction count per loop iteration of a kernel in a hipcc --save-temps ISA listing (.s): finds the backward
branches of the kernel, takes the loop with the most instructions (the time loop of the filter / smoother kernels) and
prints the instruction mix of its body.  Used for the per-wave-step figures kept under profiles/.

usage: isa_count.py file.s kernel-name-substring [--all]"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_fmac_f64_dpp") or op.startswith("v_mov_b64_dpp") or "_dpp" in op:
        return "dpp"
    if op.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64", "v_fmac_f64", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_div", "v_max_f64",
                      "v_min_f64", "v_ldexp_f64", "v_frexp", "v_cmp_", "v_trig", "v_fract", "v_rndne_f64", "v_cvt", "v_floor_f64")) and "f64" in op:
        return "valu_f64"
    if op.startswith("v_accvgpr"):
        return "agpr_mov"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, name = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = None
    for k, ln in enumerate(lines):
        if re.match(r"^[A-Za-z_]\S*:", ln) and name in ln.split(":")[0]:
            start = k
            break
    if start is None:
        sys.exit(f"kernel matching {name!r} not found")
    body = []
    for ln in lines[start + 1:]:
        if ln.startswith("\t.section") or ln.startswith(".Lfunc_end"):
            break
        body.append(ln)
    labels = {}
    headers = set()  # labels LLVM marks as loop headers
    instrs = []  # (index, op, text)
    for ln in body:
        s = ln.strip()
        if not s or s.startswith(";") or s.startswith("."):
            m = re.match(r"^(\.LBB\S+):", s)
            if m:
                labels[m.group(1)] = len(instrs)
                if "Loop Header" in s or "in Loop:" in s:  # (a rotated loop branches back to a block in front of its header)
                    headers.add(m.group(1))
            continue
        op = s.split()[0]
        instrs.append((op, s))
    loops = []
    for k, (op, s) in enumerate(instrs):
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = s.split()[1]
            if tgt in labels and labels[tgt] <= k and tgt in headers:
                loops.append((labels[tgt], k))
    if not loops:
        sys.exit("no loop found")
    loops.sort(key=lambda ab: ab[1] - ab[0], reverse=True)
    show = loops if "--all" in sys.argv else loops[:1]
    print(lines[start].split(":")[0])
    for a, b in show:
        mix = collections.Counter(classify(op) for op, _ in instrs[a:b + 1])
        total = b - a + 1
        print(f"  loop body: {total} instructions  " + "  ".join(f"{k}={v}" for k, v in sorted(mix.items(), key=lambda kv: -kv[1])))
    print(f"  whole kernel: {len(instrs)} instructions")


if __name__ == "__main__":
    main()

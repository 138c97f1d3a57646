#!/usr/bin/env python3
"""ISA check of the hand-scheduled kernels (run on the build box; hipcc cross-compiles, no GPU needed):

    python tools/isa_check.py [--out profiles/rNN_isa_check.txt]

Compiles every kernel source to gfx950 assembly (both storage-type builds), prints a per-kernel table of
VGPR / AGPR / spill / scratch figures from the code-object metadata, and FAILS (exit 1) when
  * a kernel that issues inline-asm MFMAs (the D-sliding conv / weight-gradient kernels) has a `scratch_` instruction
    between its first and last `v_mfma` - the compiler's hazard recogniser cannot see those MFMAs' operands, so a
    spill restore next to them is the hazard the hand-placed `s_nop`s do not cover (wgrad_slide.hip), or
  * in the 32- and 64-channel sliding conv kernels (all weights in AGPRs, the allocator parks loop-invariant VGPRs in the remaining
    AGPRs and a few pointers in scratch) a `v_accvgpr_write` between the MFMAs targets an AGPR that an MFMA of the
    kernel reads, or a VALU instruction writes a source register of an MFMA one or two instructions ahead of it, or
  * any kernel on the hot path spills more VGPRs than the committed allowance below (a compiler bump that pushes a
    512-register kernel over the edge shows up here, not as a silent slowdown).
"""
import argparse
import concurrent.futures as cf
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "3d-unet-renal-anatomy-extraction_amd", "csrc")
SRCS = ["conv_slide32.hip", "conv_slide64.hip", "conv_s2.hip", "conv_ws.hip", "fused_skip.hip", "wgrad_slide.hip", "wgrad_s2.hip", "conv_mfma.hip", "small_convs.hip", "conv_generic.hip",
        "norm.hip", "norm_small.hip", "loss.hip", "predict.hip", "comm.hip"]
TWICE = {"conv_slide32.hip", "conv_slide64.hip", "conv_s2.hip", "conv_ws.hip", "fused_skip.hip", "wgrad_slide.hip", "wgrad_s2.hip", "conv_mfma.hip", "small_convs.hip", "conv_generic.hip",
         "norm.hip", "norm_small.hip"}
# spilled VGPRs tolerated per kernel-name pattern (everything else: 0)
ALLOW = [(r"convt3_s2_tile_kernel", 8), (r"conv3_s1_slide32_kernel", 64), (r"conv3_s1_slide64_kernel", 64), (r"conv3_s1_pc_kernel", 40), (r"wgrad3_s1_mfma_kernel", 4), (r"wgrad3_s1_slide_kernel", 16), (r"conv3_s1_ws_kernel", 8)]
NO_SCRATCH_IN_MFMA_SPAN = [r"wgrad3_s1_slide_kernel"]
NO_COPY_INTO_MFMA_OPERANDS = [r"conv3_s1_slide32_kernel", r"conv3_s1_slide64_kernel"]


def regs_of(tok):
    """'v[4:7]' / 'a12' -> set of ('v'|'a', index)"""
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.fullmatch(r"([va])(\d+)", tok)
    return {(m.group(1), int(m.group(2)))} if m else set()


def copies_into_mfma_operands(lines, mf):
    """Two hazards the compiler cannot see through inline asm: (1) a v_accvgpr_write between the MFMAs into an AGPR
    that some MFMA of the kernel reads (the weights stay put, so any such write is a copy-back); (2) a VALU write to a
    VGPR/AGPR that one of the next two instructions, an MFMA, takes as a source."""
    tok_re = r"[va]\[\d+:\d+\]|\b[va]\d+\b"
    agpr_ops = set()
    for i in mf:
        for tok in re.findall(tok_re, lines[i]):
            agpr_ops |= {r for r in regs_of(tok) if r[0] == "a"}
    bad = 0
    code = [l for l in lines[mf[0]:mf[-1] + 1] if l.strip() and not l.strip().startswith((";", "."))]
    for n, l in enumerate(code):
        m = re.match(r"\s*v_accvgpr_write_b32\s+(a\d+)", l)
        if m and regs_of(m.group(1)) & agpr_ops:
            bad += 1
        if "v_mfma" in l:
            srcs = set()
            for tok in re.findall(tok_re, l)[1:]:
                srcs |= regs_of(tok)
            for prev in code[max(0, n - 2):n]:
                pm = re.match(r"\s*(v_\w+)\s+(" + tok_re + ")", prev)
                if pm and not pm.group(1).startswith("v_mfma") and regs_of(pm.group(2)) & srcs:
                    bad += 1
        # (3) an MFMA result read by a non-MFMA instruction fewer than 11 wait states later (MFMA = 4, s_nop k = k + 1)
        if "v_mfma" in l:
            dst = regs_of(re.findall(tok_re, l)[0])
            wait = 0
            for nxt in code[n + 1:n + 12]:
                if wait >= 11:
                    break
                if "v_mfma" not in nxt:
                    toks = re.findall(tok_re, nxt)
                    used = set()
                    for tok in toks:
                        used |= regs_of(tok)
                    if used & dst and not re.match(r"\s*s_", nxt):
                        bad += 1
                        break
                nm = re.match(r"\s*s_nop\s+(\d+)", nxt)
                wait += 4 if "v_mfma" in nxt else (int(nm.group(1)) + 1 if nm else 1)
    return bad


def compile_asm(src, f16, tmp):
    out = os.path.join(tmp, src.replace(".hip", "_f16.s" if f16 else ".s"))
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S",
           *(["-fno-slp-vectorize"] if src in ("conv_slide32.hip", "conv_slide64.hip") else []),   # as the Makefile
           "-Wno-unused-function", "-Wno-pass-failed", os.path.join(CSRC, src), "-o", out]
    if f16:
        cmd.insert(1, "-DRU3D_STORAGE_F16")
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def demangle(names):
    try:
        p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True,
                           text=True, check=True)
        return dict(zip(names, p.stdout.splitlines()))
    except Exception:
        return {n: n for n in names}


def parse(path):
    text = open(path).read()
    kernels = {}
    # bodies: from "<name>:" to the matching ".Lfunc_end"
    for m in re.finditer(r"^(\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if ".amdhsa_kernel " + name not in text:
            continue
        lines = body.splitlines()
        mf = [i for i, l in enumerate(lines) if "v_mfma" in l]
        inside = 0
        if mf:
            inside = sum(1 for l in lines[mf[0]:mf[-1] + 1] if re.search(r"\bscratch_(load|store)", l))
        kernels[name] = {"mfma": len(mf), "scratch_in_mfma_span": inside,
                         "copies_into_operands": copies_into_mfma_operands(lines, mf) if mf else 0,
                         "scratch_total": sum(1 for l in lines if re.search(r"\bscratch_(load|store)", l))}
    for m in re.finditer(r"- \.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?"
                         r"\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", text, re.S):
        agpr, name, scratch, sgpr, vgpr, spill = m.groups()
        if name in kernels:
            kernels[name].update(agpr=int(agpr), scratch_bytes=int(scratch), sgpr=int(sgpr), vgpr=int(vgpr),
                                 spill=int(spill))
    return kernels


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    jobs = [(s, False) for s in SRCS] + [(s, True) for s in SRCS if s in TWICE]
    rows, failures = [], []
    with tempfile.TemporaryDirectory() as tmp:
        with cf.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
            paths = list(ex.map(lambda j: compile_asm(j[0], j[1], tmp), jobs))
        for (src, f16), path in zip(jobs, paths):
            ks = parse(path)
            names = demangle(list(ks))
            for raw, k in sorted(ks.items(), key=lambda kv: names[kv[0]]):
                nice = names[raw]
                nice = re.sub(r"\(anonymous namespace\)::|ru3d_bf16::|ru3d_f16::|void ", "", nice)
                nice = re.sub(r"\(.*$", "", nice)
                allow = max([a for pat, a in ALLOW if re.search(pat, nice)] or [0])
                rows.append((src + (" [f16]" if f16 else ""), nice, k.get("vgpr", -1), k.get("agpr", -1),
                             k.get("sgpr", -1), k.get("spill", -1), k.get("scratch_bytes", -1), k["mfma"],
                             k["scratch_in_mfma_span"]))
                if k.get("spill", 0) > allow:
                    failures.append("%s: %s spills %d VGPRs (allowance %d)" % (src, nice, k["spill"], allow))
                if any(re.search(p, nice) for p in NO_SCRATCH_IN_MFMA_SPAN) and k["scratch_in_mfma_span"]:
                    failures.append("%s: %s has %d scratch instruction(s) between its first and last v_mfma"
                                    % (src, nice, k["scratch_in_mfma_span"]))
                if any(re.search(p, nice) for p in NO_COPY_INTO_MFMA_OPERANDS) and k["copies_into_operands"]:
                    failures.append("%s: %s copies into an MFMA operand register next to / between its MFMAs "
                                    "(%d times)" % (src, nice, k["copies_into_operands"]))
    lines = ["%-24s %-62s %5s %5s %5s %6s %8s %6s %s" % ("source", "kernel", "vgpr", "agpr", "sgpr", "spill", "scratchB",
                                                          "mfma", "scratch-in-mfma-span")]
    for r in rows:
        lines.append("%-24s %-62s %5d %5d %5d %6d %8d %6d %d" % (r[0], r[1][:62], *r[2:]))
    lines.append("")
    lines.append("FAILURES: %d" % len(failures))
    lines += failures
    text = "\n".join(lines)
    print(text)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text + "\n")
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()

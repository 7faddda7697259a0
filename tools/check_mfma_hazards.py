#!/usr/bin/env python3
"""Scan hipcc's gfx950 assembly for MFMA results that are read across a basic-block boundary too soon.

hipcc's hazard recognizer pads 'XDL write VGPR -> VALU / VMEM / LDS read' inside a basic block, but (ROCm 7.2) an MFMA that is the last
vector instruction before a branch or a fall-through label is not padded against the first readers in the successor block: found in
attn_varlen_kernel (the tile maximum read the score registers of a 16x16x4 f32 MFMA before they were written: harmless there -- any shift
works in a softmax -- but non-deterministic).  For every v_mfma this script walks forward through fall-through AND taken branches for
WAIT wait states and reports any instruction that reads a register of the MFMA's destination.
Usage: python tools/check_mfma_hazards.py file.s [...]      (hipcc -S --cuda-device-only ...)"""
import re, sys

# passes (4 cycles each) on gfx950: FLOPs of the instruction / (FLOPs per SIMD and cycle of its type) / 4
PASSES = {"16x16x4_f32": 8, "32x32x2_f32": 16, "16x16x32_f16": 4, "16x16x32_bf16": 4, "32x32x16_f16": 8, "32x32x16_bf16": 8, "16x16x4_f64": 16,
          "16x16x16_f16": 8, "16x16x16_bf16": 8, "32x32x8_f16": 16, "32x32x8_bf16": 16}    # the K-halved forms: at most these (conservative)
def need(op):
    for k, p in PASSES.items():
        if k in op:
            return p
    return 16
REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")
def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out |= {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
        else:
            out.add((m.group(4), int(m.group(5))))
    return out
def states(ins):
    m = re.match(r"s_nop (\d+)", ins)
    return int(m.group(1)) + 1 if m else 1

def scan(path):
    lines = [l.split(";")[0].rstrip() for l in open(path)]
    label = {l[:-1]: i for i, l in enumerate(lines) if re.match(r"^[.\w$]+:$", l)}
    fn = None
    bad = 0
    for i, l in enumerate(lines):
        if re.match(r"^_Z\w+:$", l): fn = l[:-1]
        t = l.strip()
        if not t.startswith("v_mfma"): continue
        ops = t.split(None, 1)[1].split(",")
        dst = regs(ops[0])
        P = need(t.split()[0])
        W = P + 2
        seen = set()
        work = [(i + 1, 0)]
        while work:
            j, w = work.pop()
            while j < len(lines) and w < W:
                if (j, w) in seen: break
                seen.add((j, w))
                u = lines[j].strip()
                if not u or u.startswith(".") or u.endswith(":") or u.startswith(";"):
                    j += 1; continue
                op = u.split()[0]
                if op == "s_endpgm": break
                if op.startswith("s_cbranch") or op == "s_branch":
                    tgt = u.split()[-1]
                    if tgt in label: work.append((label[tgt], w + 1))
                    if op == "s_branch": break
                    j += 1; w += 1; continue
                if op.startswith("v_mfma"):
                    # a dependent MFMA (SrcC = previous vdst) is interlocked by the hardware; an overwrite ends the window
                    if regs(u.split(None, 1)[1].split(",")[0]) & dst: break
                elif op[0] in "vdgbs" and not op.startswith("s_"):
                    body = u.split(None, 1)[1] if " " in u else ""
                    parts = body.split(",")
                    is_store = op.startswith(("global_store", "buffer_store", "ds_write", "scratch_store", "flat_store", "ds_store"))
                    src = ",".join(parts if is_store else parts[1:])
                    req = P + 2      # what hipcc itself pads to inside a basic block (s_nop 9 behind an 8-pass MFMA, 17 wait states behind a 16-pass one)
                    if regs(src) & dst and w >= req: break
                    if regs(src) & dst:
                        W = req
                        print(f"{path}:{j + 1}: {fn}: `{u}` reads the result of `{t}` (line {i + 1}) after {w} wait states, {W} needed")
                        bad += 1
                        break
                    if regs(parts[0]) & dst and not is_store: break          # overwritten
                w += states(u); j += 1
    return bad

if __name__ == "__main__":
    total = sum(scan(p) for p in sys.argv[1:])
    print("violations:", total)
    sys.exit(1 if total else 0)

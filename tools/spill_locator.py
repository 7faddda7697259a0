#!/usr/bin/env python3
"""Where do a kernel's spills sit?  Reads hipcc's ISA listing (--save-temps, ideally with -gline-tables-only for .loc lines) and, per
kernel, lists every scratch_store / scratch_load with the loop nest around it (loops = backward branches to an earlier label) and
the source line it belongs to.  A spill in the prologue or a one-off phase costs a few cycles per workgroup; a reload inside a loop is
a vmcnt(0) on the loop's critical path.

usage: python tools/spill_locator.py <listing.s> <kernel-name-substring> [...]     (substring of the MANGLED name, e.g. Li1024ELi28ELi2ELb0ELb1E)
       python tools/spill_locator.py --build sort.hip <substring> ...             (compiles fusion_amd/csrc/<file> into /tmp/isa first: source lines)
       python tools/spill_locator.py --object sort.o <substring> ...              (disassembles the SHIPPED object: seconds, no source lines)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = "-O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -Wno-unused-function".split()


def build(src, out_dir="/tmp/isa"):
    os.makedirs(out_dir, exist_ok=True)
    stem = os.path.splitext(os.path.basename(src))[0]
    subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "-gline-tables-only", "-save-temps", "-c", os.path.join(ROOT, "fusion_amd", "csrc", src),
                           "-o", os.path.join(out_dir, stem + ".o")], cwd=out_dir, stderr=subprocess.DEVNULL)
    return os.path.join(out_dir, f"{stem}-hip-amdgcn-amd-amdhsa-gfx950.s")


def disassemble(obj, out_dir="/tmp/isa_obj"):
    """The gfx950 code object inside a built .o (the shipped build: fusion_amd/csrc/<name>.o), disassembled with symbolised branch
    targets -- seconds, no recompilation, no source lines.  Returns the listing's path."""
    os.makedirs(out_dir, exist_ok=True)
    stem = os.path.splitext(os.path.basename(obj))[0]
    fat, co, dis = (os.path.join(out_dir, stem + ext) for ext in (".fatbin", ".co", ".dis"))
    subprocess.check_call(["objcopy", "--dump-section", f".hip_fatbin={fat}", obj])
    subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           f"--input={fat}", f"--output={co}", "--unbundle"])
    with open(dis, "w") as f:
        subprocess.check_call(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--symbolize-operands", co], stdout=f)
    return dis


def kernels(listing):
    """{mangled name: [lines]} for every kernel body.  Two input forms: hipcc's assembly listing (--save-temps: label .. .Lfunc_end)
    and llvm-objdump's disassembly of a code object (`<name>:` headers, `<L12>:` labels, rewritten here to the listing's .LBB form)."""
    if listing.endswith(".dis"):
        out, cur, fn = {}, None, 0
        for line in open(listing, errors="replace"):
            m = re.match(r"^[0-9a-f]+ <(\w+)>:", line)
            if m and re.fullmatch(r"L\d+", m.group(1)):           # a branch target inside the current function
                if cur is not None:
                    cur.append(f".LBB{fn}_{m.group(1)[1:]}:")
                continue
            if m:
                cur = out.setdefault(m.group(1), []); fn += 1
                continue
            if cur is None:
                continue
            ins = line.split("//")[0].strip()
            if ins:
                cur.append(re.sub(r"\bL(\d+)\b", lambda mm: f".LBB{fn}_{mm.group(1)}", ins) if ins.startswith(("s_cbranch", "s_branch")) else ins)
        return out
    out, cur, name = {}, None, None
    for line in open(listing, errors="replace"):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m and cur is None:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            cur.append(line.rstrip("\n"))
            if line.startswith(".Lfunc_end"):
                out[name] = cur
                cur = None
    return out


def analyse(body):
    """-> (scratch accesses, #instructions, loops).  Loops are found on the control-flow graph: basic blocks, edges (fall-through +
    branch targets), strongly connected components -- a backward branch alone may be a cold block laid out at the end of the function
    jumping back into the main flow, not a loop.  An access is "inside a loop" when its block lies on a cycle; the innermost loop
    reported is the shortest backward-branch interval around it whose two ends lie on that same cycle."""
    labels, instrs, loc = {}, [], None
    for ln in body:
        s = ln.strip()
        m = re.match(r"^\.loc\s+\d+\s+(\d+)", s)
        if m:
            loc = int(m.group(1)); continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            labels[m.group(1)] = len(instrs); continue
        if not s or s.startswith((";", ".", "//")):
            continue
        instrs.append((s.split(";")[0].strip(), loc))
    n = len(instrs)
    # basic blocks
    leaders = {0} | set(labels.values())
    br = {}
    for i, (ins, _) in enumerate(instrs):
        m = re.match(r"^(s_c?branch\w*)\s+(\.LBB\d+_\d+)", ins)
        if m:
            br[i] = (m.group(1), labels.get(m.group(2)))
            leaders.add(i + 1)
        elif ins.startswith(("s_endpgm", "s_setpc")):
            leaders.add(i + 1)
    starts = sorted(x for x in leaders if x < n)
    block_of = [0] * n
    for b, st in enumerate(starts):
        for i in range(st, starts[b + 1] if b + 1 < len(starts) else n):
            block_of[i] = b
    succ = [[] for _ in starts]
    for b, st in enumerate(starts):
        last = (starts[b + 1] if b + 1 < len(starts) else n) - 1
        ins = instrs[last][0]
        if last in br:
            kind, tgt = br[last]
            if tgt is not None:
                succ[b].append(block_of[tgt])
            if kind != "s_branch" and last + 1 < n:
                succ[b].append(block_of[last + 1])
        elif not ins.startswith(("s_endpgm", "s_setpc")) and last + 1 < n:
            succ[b].append(block_of[last + 1])
    # Tarjan, iterative
    idx, low, comp, onst, st_, cnt, ncomp = {}, {}, {}, set(), [], [0], [0]
    for root in range(len(starts)):
        if root in idx:
            continue
        work = [(root, 0)]
        while work:
            v, pi = work[-1]
            if pi == 0:
                idx[v] = low[v] = cnt[0]; cnt[0] += 1; st_.append(v); onst.add(v)
            rec = False
            for k in range(pi, len(succ[v])):
                w = succ[v][k]
                if w not in idx:
                    work[-1] = (v, k + 1); work.append((w, 0)); rec = True; break
                if w in onst:
                    low[v] = min(low[v], idx[w])
            if rec:
                continue
            if low[v] == idx[v]:
                while True:
                    w = st_.pop(); onst.discard(w); comp[w] = ncomp[0]
                    if w == v:
                        break
                ncomp[0] += 1
            work.pop()
            if work:
                u = work[-1][0]
                low[u] = min(low[u], low[v])
    size = {}
    for b, c in comp.items():
        size[c] = size.get(c, 0) + 1
    cyc = lambda b: size[comp[b]] > 1 or b in succ[b]
    inv = {v: k for k, v in labels.items()}
    loops = [(tgt, i, inv.get(tgt, "?")) for i, (_, tgt) in br.items()
             if tgt is not None and tgt <= i and comp[block_of[tgt]] == comp[block_of[i]] and cyc(block_of[i])]
    found = []
    for i, (ins, src) in enumerate(instrs):
        if ins.startswith(("scratch_store", "scratch_load")):
            b = block_of[i]
            around = sorted((l for l in loops if l[0] <= i <= l[1] and comp[block_of[l[0]]] == comp[b]), key=lambda l: l[1] - l[0]) if cyc(b) else []
            found.append(dict(idx=i, op=ins.split()[0], depth=len(around), loop=around[0] if around else None, src=src))
    return found, n, loops


def report(listing, pats):
    ks = kernels(listing)
    res = {}
    for name, body in ks.items():
        if not any(p in name for p in pats):
            continue
        found, n, loops = analyse(body)
        res[name] = found
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        inloop = [f for f in found if f["depth"] > 0]
        print(f"== {dem}\n   {n} instructions, {len(loops)} loops, {len(found)} scratch accesses "
              f"({sum(f['op'].startswith('scratch_store') for f in found)} stores, {sum(f['op'].startswith('scratch_load') for f in found)} loads), "
              f"{len(inloop)} inside a loop")
        by = {}
        for f in found:
            key = (f["src"], f["loop"][2] if f["loop"] else "-", (f["loop"][1] - f["loop"][0]) if f["loop"] else 0)
            by.setdefault(key, [0, 0])[0 if f["op"].startswith("scratch_store") else 1] += 1
        for (src, lab, span), (st, ld) in sorted(by.items(), key=lambda kv: (kv[0][0] or 0, kv[0][1])):
            where = "straight-line code" if lab == "-" else f"loop {lab} ({span} instructions long)"
            print(f"   source line {src}: {st} stores, {ld} loads in {where}")
    return res


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--build":
        listing = build(args[1]); pats = args[2:]
    elif args and args[0] == "--object":
        listing = disassemble(os.path.join(ROOT, "fusion_amd", "csrc", args[1])); pats = args[2:]
    else:
        listing, pats = args[0], args[1:]
    report(listing, pats)

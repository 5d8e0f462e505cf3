#!/usr/bin/env python3
"""Static audit of the f32-A ping-pong GEMM (csrc/gemm_pp.hip, uic_gemm_pp_kernel<RT, false, true>): its A units are loaded by
inline-asm `global_load_dwordx4` into C++ variables and waited for by hand-counted `s_waitcnt vmcnt(N)` four phases later.  hipcc
treats an asm statement's VGPR destination as written when the statement ends (cdna_hip_programming.md 5.7 item 1), so between
the load and the commit nothing but the kernel's own conversions may touch those registers: a compiler-inserted copy or spill
there would read registers whose data has not landed.  This script compiles the file with -save-temps (or takes an existing .s)
and checks, for every f32-A instantiation:
  * no scratch (spill) memory at all, and .vgpr_spill_count 0;
  * between an asm load and the `v_cvt_pk_bf16_f32` that consumes its destination, no instruction outside an asm block reads or
    writes that destination.
Exit status 0 = clean.  tests/test_host_logic.py runs it (no GPU needed: hipcc cross-compiles)."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "unpaired_image_captioning_amd", "csrc", "gemm_pp.hip")
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def _blocks(lines):
    """Basic blocks of one kernel's listing: [(label or None, [(line number, text, in_asm)], successors)]; successors are labels or
    the index of the fall-through block."""
    blocks = []
    cur = {"label": None, "ins": [], "succ": []}
    in_asm = False
    for ln, raw in lines:
        t = raw.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        line = raw.split(";")[0].strip()
        if not line or line.startswith(".") and not line.endswith(":"):
            continue
        if line.endswith(":"):
            if cur["ins"] or cur["label"] is not None:
                cur["succ"].append("FALL")
                blocks.append(cur)
            cur = {"label": line[:-1], "ins": [], "succ": []}
            continue
        cur["ins"].append((ln, line, in_asm))
        op = line.split()[0]
        if op == "s_branch" or op.startswith("s_cbranch") or op == "s_endpgm":
            if op != "s_endpgm":
                cur["succ"].append(line.split()[1])
            if op.startswith("s_cbranch"):
                cur["succ"].append("FALL")
            blocks.append(cur)
            cur = {"label": None, "ins": [], "succ": []}
    if cur["ins"]:
        blocks.append(cur)
    index = {b["label"]: i for i, b in enumerate(blocks) if b["label"] is not None}
    for i, b in enumerate(blocks):
        b["next"] = [index[x] if x != "FALL" else i + 1 for x in b["succ"] if x == "FALL" and i + 1 < len(blocks) or x in index]
    return blocks


def audit_kernel(name, lines):
    """Forward analysis over the kernel's control-flow graph: a register is "in flight" from the asm load that writes it to the
    v_cvt_pk_bf16_f32 that reads it (on every path into a block); any other instruction outside an asm block that touches an
    in-flight register is a problem.  Problems are collected on the final (fixpoint) pass only."""
    blocks = _blocks(lines)
    problems = {}
    n_loads = n_cvt = 0
    entry = [None] * len(blocks)
    entry[0] = {}
    work = [0]
    counted = set()
    while work:
        i = work.pop()
        inflight = dict(entry[i])
        for ln, _, _ in blocks[i]["ins"]:
            problems.pop(ln, None)
        for ln, line, in_asm in blocks[i]["ins"]:
            op = line.split()[0]
            if "scratch_" in op:
                problems[ln] = "%s: line %d uses scratch memory: %s" % (name, ln, line)
            if in_asm:
                if op.startswith("global_load_dwordx4"):
                    for r in regs_of(line.split(None, 1)[1].split(",")[0]):
                        inflight[r] = ln
                    if ln not in counted:
                        counted.add(ln)
                        n_loads += 1
                continue
            touched = regs_of(line.split(None, 1)[1]) if " " in line else set()
            hit = touched & set(inflight)
            if not hit:
                continue
            if op.startswith("v_cvt_pk_bf16_f32"):
                for r in regs_of(",".join(line.split(None, 1)[1].split(",")[1:])) & set(inflight):
                    del inflight[r]
                if ln not in counted:
                    counted.add(ln)
                    n_cvt += 1
                continue
            problems[ln] = "%s: line %d touches v%s while its asm load (line %d) has not been committed: %s" % (
                name, ln, sorted(hit), inflight[sorted(hit)[0]], line)
        for j in blocks[i]["next"]:
            if entry[j] is None:
                entry[j] = dict(inflight)
                work.append(j)
            else:
                # MUST analysis (intersection at merges): loads and commits that sit under the same wave-uniform condition (`two`,
                # the image stores) would otherwise be carried along paths no wave takes
                keep = {r: l for r, l in entry[j].items() if r in inflight}
                if len(keep) != len(entry[j]):
                    entry[j] = keep
                    work.append(j)
    # the K loop ends at the kernel's last s_barrier; behind it (the epilogue) every load has been committed -- what the
    # path-insensitive analysis still carries there are loads and commits that sit under the same wave-uniform condition
    last_barrier = max([ln for b in blocks for ln, line, _ in b["ins"] if line.split()[0] == "s_barrier"] or [0])
    out = [problems[k] for k in sorted(problems) if k <= last_barrier or "scratch" in problems[k]]
    if n_loads == 0 or n_cvt == 0:
        out.append("%s: found %d asm loads and %d conversions -- the audit no longer recognises the kernel" % (name, n_loads, n_cvt))
    return out, n_loads, n_cvt


def main():
    tmp = None
    if len(sys.argv) > 1:
        spath = sys.argv[1]
    else:
        tmp = tempfile.mkdtemp(prefix="uic_audit_")
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-save-temps", "-c", SRC, "-o", os.path.join(tmp, "pp.o")],
                           cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            shutil.rmtree(tmp, ignore_errors=True)
            sys.exit("audit_f32a_asm: %s failed (%d):\n%s" % (hipcc, r.returncode, r.stdout[-4000:]))
        spath = [os.path.join(tmp, f) for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
    text = open(spath).read().split("\n")
    if tmp is not None:
        shutil.rmtree(tmp, ignore_errors=True)
    start = re.compile(r"^(_ZN12_GLOBAL__N_118uic_gemm_pp_kernelILi(\d)ELb0ELb1EEEv13UicGemmParams):")
    kernels = []
    cur = None
    for i, l in enumerate(text):
        m = start.match(l)
        if m:
            cur = (m.group(1), [])
            kernels.append(cur)
            continue
        if cur is not None:
            cur[1].append((i + 1, l))
            if l.strip().startswith("s_endpgm"):
                cur = None
    bad = []
    if len(kernels) < 2:
        bad.append("expected the 192- and 128-row f32-A instantiations, found %d" % len(kernels))
    for name, lines in kernels:
        p, nl, nc = audit_kernel(name, lines)
        print("%s: %d asm loads, %d conversions, %d problems" % (name, nl, nc, len(p)))
        bad += p
    for m in re.finditer(r"\.name:\s+(_ZN12_GLOBAL__N_118uic_gemm_pp_kernelILi\dELb0ELb1EEEv13UicGemmParams)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", "\n".join(text)):
        if int(m.group(2)) != 0:
            bad.append("%s: .vgpr_spill_count %s" % (m.group(1), m.group(2)))
    for b in bad:
        print("PROBLEM", b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

"""Audit of the hand-register-allocated kernels (csrc/attention_bwd64.h): the compiler is fenced into v0..v63 and owns no AGPR there; every register above
is named literally inside `asm volatile` statements.  This script compiles attention.hip to ISA and checks, for every *64w kernel, that no
COMPILER-GENERATED instruction (anything outside ;;#ASMSTART ... ;;#ASMEND) names a VGPR above v63 or any AGPR, that nothing was spilled, and that the
loop bodies contain no scalar memory loads (they would break the counted lgkmcnt waits).

    python tools/audit_asm_owned.py            # exit code 0 = clean
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "visper-lm_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def isa(debug=False):
    """attention.hip -> ISA with the Makefile's flags: the sealed product build (default) or the -DVP_DEBUG build (debug=True)."""
    out = os.path.join(tempfile.mkdtemp(prefix="vp_audit_"), "attention.s")
    cmd = [HIPCC] + (["-DVP_DEBUG"] if debug else []) + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffast-math", "-fno-finite-math-only", "-mllvm",
           "-amdgpu-spill-vgpr-to-agpr=0", "--cuda-device-only", "-S", os.path.join(CSRC, "attention.hip"), "-o", out]
    subprocess.run(cmd, check=True, cwd=CSRC, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(out).read()


def audit(text):
    """AGPRs: never outside asm.  VGPRs above v63: not between the first asm statement that WRITES one of them (a literal register goes live) and the
    LAST asm statement of the kernel that names an owned register (the epilogue's RoPE-table loads into v64..v191, its v_mov reads of them and the
    v_accvgpr reads of the accumulators: round 5's window ended at the last barrier and left the epilogue unchecked)."""
    problems, seen = [], 0
    for m in re.finditer(r"^(_Z\d+attn_bwd_(?:dq|dkdv)64w_kernel\w+):.*?^\s*s_endpgm", text, re.S | re.M):
        name, lines = m.group(1), m.group(0).splitlines()
        seen += 1
        in_asm, live, last_bar, first_bar, last_owned = False, None, 0, None, 0
        for ln, line in enumerate(lines):
            t = line.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
            elif t.startswith(";;#ASMEND"):
                in_asm = False
            elif in_asm:
                if t.startswith("s_barrier"):
                    last_bar = ln
                    first_bar = ln if first_bar is None else first_bar
                if re.search(r"\bv(6[4-9]|[7-9]\d|1\d\d|2\d\d)\b|\bv\[(6[4-9]|[7-9]\d|1\d\d|2\d\d):|\ba\d+\b|\ba\[\d+:", t.split(";")[0]):
                    last_owned = ln
                if live is None and re.match(r"(v_mov_b32 v(6[4-9]|[7-9]\d|1\d\d|2\d\d)\b|ds_read\w* v\[(6[4-9]|[7-9]\d|1\d\d|2\d\d):)", t):
                    live = ln
        in_asm = False
        for ln, line in enumerate(lines):
            t = line.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if in_asm or not t or t.startswith((";", ".", "_Z")) or t.endswith(":"):
                continue
            code = t.split(";")[0]
            if re.search(r"\ba\d+\b|\ba\[\d+:\d+\]|v_accvgpr", code):
                problems.append(f"{name}:{ln}: compiler instruction touches an AGPR: {code}")
            if code.startswith("scratch_") and first_bar is not None and first_bar < ln < last_bar:
                problems.append(f"{name}:{ln}: scratch access between the barriers (the streams' loops): {code}")
            if live is not None and live <= ln <= max(last_bar, last_owned):
                hi = [int(r) for r in re.findall(r"\bv(\d+)\b", code)] + [int(b_) for _, b_ in re.findall(r"\bv\[(\d+):(\d+)\]", code)]
                if hi and max(hi) > 63:
                    problems.append(f"{name}:{ln}: compiler instruction uses v{max(hi)} while asm-owned registers are live: {code}")
                if first_bar is not None and ln > first_bar and code.startswith(("s_load_", "s_buffer_load")):
                    problems.append(f"{name}:{ln}: scalar memory load inside the counted-lgkmcnt region: {code}")
    return seen, problems


if __name__ == "__main__":
    rc = 0
    for dbg in (False, True):
        n, probs = audit(isa(debug=dbg))
        print(f"{'-DVP_DEBUG' if dbg else 'sealed'} build: {n} kernels audited, {len(probs)} problems")
        for p_ in probs[:40]:
            print("  ", p_)
        rc |= 1 if probs or n == 0 else 0
    sys.exit(rc)

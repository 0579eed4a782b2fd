"""Static audit of the one-wave-per-SIMD GEMM's epilogues for the hazard behind round 4's "NaNs beside other kernels" (gemm.hip W4_KEEP2): an
`asm volatile("v_accvgpr_read_b32 vN, aM")` — invisible to the compiler's hazard recogniser — must not write a VGPR that one of the most recent
buffer_store instructions reads as DATA.  On gfx950 nothing orders such a read behind the store's (late) data fetch when the CU's memory pipeline is
shared with another kernel's waves; the store then writes the next block's raw fp32 accumulator (tools/nan_pattern_r06.py).

Compiles gemm.hip to ISA with the Makefile's flags and checks every gemm_nt_256w4 instantiation: for each v_accvgpr_read inside an ASM block, its
destination must not be a data register of any of the last WINDOW buffer_store instructions issued since the last `s_waitcnt vmcnt(0)`.

    python tools/w4_store_data_audit.py            # exit code 0 = clean
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "visper-lm_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
WINDOW = 0   # (unused: blocks are delimited by the sched_barrier markers)            # stores whose data registers stay protected: those of the previous 16-row block (two 16-byte stores + the sum-of-squares dword: what W4_KEEP2 keeps live)


def isa(extra=()):
    out = os.path.join(tempfile.mkdtemp(prefix="vp_w4audit_"), "gemm.s")
    cmd = [HIPCC, *extra, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffast-math", "-fno-finite-math-only", "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0",
           "--cuda-device-only", "-S", os.path.join(CSRC, "gemm.hip"), "-o", out]
    subprocess.run(cmd, check=True, cwd=CSRC, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(out).read()


def _regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def audit(text):
    """Protected = the data registers of the last two 16-byte stores (one 16-row block of one 64-column half: what W4_KEEP2 keeps live) and of any
    narrower store issued since.  Labels reset the state (the listing order across basic blocks is not the execution order; the unrolled epilogue
    loops have no label inside)."""
    res = {}
    for m in re.finditer(r"^(_Z\d+gemm_nt_256w4\w+):.*?^\s*s_endpgm", text, re.S | re.M):
        name, recent, in_asm, probs, n_reads, n_stores = m.group(1), [], False, [], 0, 0
        for ln, line in enumerate(m.group(0).splitlines()):
            t = line.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            code = t.split(";")[0].strip()
            if code.endswith(":"):
                recent = []
            if not code or code.startswith(".") or code.endswith(":"):
                continue
            if code.startswith("buffer_store_dword"):
                n_stores += 1
                wide = code.startswith("buffer_store_dwordx4")
                recent.append((ln, _regs(code.split()[1].rstrip(",")), wide))
                wides = [k for k, r in enumerate(recent) if r[2]]
                if len(wides) > 2:
                    recent = recent[wides[-2]:]
            elif code.startswith("s_waitcnt") and "vmcnt(0)" in code:
                recent = []
            elif in_asm and code.startswith("v_accvgpr_read_b32"):
                n_reads += 1
                dst = _regs(code.split()[1].rstrip(","))
                for sl, data, _ in recent:
                    if dst & data:
                        probs.append(f"{name}:{ln}: `{code}` overwrites a data register of the buffer_store at line {sl}")
        res[name] = (n_reads, n_stores, probs)
    return res


if __name__ == "__main__":
    bad = 0
    for name, (nr, ns, probs) in audit(isa(sys.argv[1:])).items():
        print(f"{name}: {nr} asm accumulator reads, {ns} buffer stores, {len(probs)} overwrite a recent store's data register")
        for p_ in probs[:6]:
            print("   ", p_)
        bad += len(probs)
    sys.exit(1 if bad else 0)

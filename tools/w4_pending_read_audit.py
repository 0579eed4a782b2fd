"""Dev tool (CPU): does any instruction of a gemm_nt_256w4 instantiation touch a VGPR that an asm `ds_read` has written and no `s_waitcnt lgkmcnt(0)`
has covered yet?  The hand-scheduled K loop issues its fragment reads as asm statements, which are invisible to the compiler's wait-count
bookkeeping: a compiler-generated copy / spill / use of such a register in front of the wait reads whatever the register held before.

    python tools/w4_pending_read_audit.py file.s [kernel-substring]
"""
import re
import sys


def regs(tok):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(a), int(b) + 1))
    out.update(int(r) for r in re.findall(r"\bv(\d+)\b", tok))
    return out


def audit(text, want):
    res = {}
    for m in re.finditer(r"^(_Z\d+gemm_nt_256w4\w+):.*?^\s*s_endpgm", text, re.S | re.M):
        name = m.group(1)
        if want and want not in name:
            continue
        pending, in_asm, probs, n_ds = {}, False, [], 0
        for ln, line in enumerate(m.group(0).splitlines()):
            t = line.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not t or t.startswith((";", ".")) or t.endswith(":"):
                continue
            code = t.split(";")[0].strip()
            if code.startswith("s_waitcnt"):
                mm = re.search(r"lgkmcnt\((\d+)\)", code)
                if mm and int(mm.group(1)) == 0:
                    pending.clear()
                elif mm is None and "vmcnt" not in code and "expcnt" not in code:       # bare s_waitcnt 0
                    pending.clear()
                continue
            if code.startswith("ds_read") and in_asm:
                dst = code.split(",")[0]
                for r in regs(dst):
                    pending[r] = ln
                n_ds += 1
                continue
            touched = regs(code) & set(pending)
            if touched:
                probs.append((ln, "asm" if in_asm else "COMPILER", code, sorted(touched)[:4], min(pending[r] for r in touched)))
        res[name] = (n_ds, probs)
    return res


if __name__ == "__main__":
    r = audit(open(sys.argv[1]).read(), sys.argv[2] if len(sys.argv) > 2 else "")
    for name, (n_ds, probs) in r.items():
        comp = [p for p in probs if p[1] == "COMPILER"]
        print(f"{name}: {n_ds} asm ds_reads, {len(probs)} instructions touch a register with a read in flight ({len(comp)} compiler-generated)")
        for p in probs[:12]:
            print(f"    line {p[0]} [{p[1]}] {p[2]}   (regs {p[3]}, read issued at line {p[4]})")

"""Per-kernel register / scratch figures of the built library, read from the code objects' own metadata (no GPU needed).

    python tools/kernel_resources.py [visper-lm_amd/libvisper_hip.so] [name-filter]

The library is a host ELF with uncompressed clang offload bundles inside; every bundle entry for gfx950 is an AMDGPU ELF whose NT_AMDGPU_METADATA note
lists .name / .vgpr_count / .agpr_count / .sgpr_count / .vgpr_spill_count / .private_segment_fixed_size per kernel.  Used by tests/test_build_resources.py
to pin what hand-scheduled kernels assume about their own allocation (the 4-wave GEMM must own its SIMD's whole register file)."""
import os
import re
import struct
import subprocess
import sys
import tempfile

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_objects(path):
    data = open(path, "rb").read()
    out = []
    for m in re.finditer(MAGIC, data):
        base = m.start()
        (n,) = struct.unpack_from("<Q", data, base + len(MAGIC))
        p = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx" in triple and size > 0:
                out.append((triple, data[base + off:base + off + size]))
    return out


def kernels(path):
    """{kernel name (demangled when c++filt is there): dict of the integer metadata fields}"""
    res = {}
    for triple, blob in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        cur = None
        for line in txt.splitlines():
            m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
            if not m:
                continue
            k, v = m.group(1), m.group(2).strip()
            if k == "agpr_count":                                   # first field of a kernel's map (alphabetical)
                cur = {}
            if cur is None:
                continue
            if k == "name":
                res[v.strip("'\"")] = cur
            elif re.fullmatch(r"-?\d+", v):
                cur[k] = int(v)
    try:
        names = list(res)
        dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
        if len(dem) == len(names):
            res = {d: res[n] for d, n in zip(dem, names)}
    except OSError:
        pass
    return res


if __name__ == "__main__":
    so = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else os.path.join(os.path.dirname(__file__), "..", "visper-lm_amd", "libvisper_hip.so")
    flt = sys.argv[2] if len(sys.argv) > 2 else (sys.argv[1] if len(sys.argv) > 1 and not os.path.exists(sys.argv[1]) else "")
    for name, d in sorted(kernels(so).items()):
        if flt in name:
            print(f"{d.get('vgpr_count', -1):4d} v {d.get('agpr_count', -1):4d} a {d.get('sgpr_count', -1):4d} s  spill {d.get('vgpr_spill_count', 0):3d}  "
                  f"scratch {d.get('private_segment_fixed_size', 0):5d}  lds {d.get('group_segment_fixed_size', 0):6d}  {name[:150]}")

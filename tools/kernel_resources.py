"""Registers, scratch and LDS of the kernels in the built library's gfx950 code object (no GPU needed).

usage: python tools/kernel_resources.py [substring of the mangled kernel name, default k_arcte_lines]
"""
import os
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "reveal-graph-embedding_amd", "csrc", "libarcte_hip.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_object(path):
    b = open(path, "rb").read()
    i = b.find(b"__CLANG_OFFLOAD_BUNDLE__")
    n = struct.unpack_from("<Q", b, i + 24)[0]
    off = i + 32
    for _ in range(n):
        o, sz, tl = struct.unpack_from("<QQQ", b, off)
        off += 24
        t = b[off:off + tl]
        off += tl
        if b"gfx950" in t:
            return b[i + o:i + o + sz]
    raise SystemExit("no gfx950 code object in " + path)


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else "k_arcte_lines"
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(code_object(SO))
        f.flush()
        notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
    cur = {}
    rows = []
    for line in notes.splitlines():
        line = line.strip()
        if line.startswith("- .agpr_count:") or line.startswith("- .args:"):
            if cur.get("name"):
                rows.append(cur)
            cur = {}
        for key in (".name:", ".vgpr_count:", ".sgpr_count:", ".private_segment_fixed_size:", ".agpr_count:", ".group_segment_fixed_size:"):
            if key in line:
                cur[key.strip(".:")] = line.split(key)[1].strip()
    if cur.get("name"):
        rows.append(cur)
    total = 0
    for r in rows:
        total += 1
        if want in r.get("name", ""):
            demangled = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
            print("%-110s vgpr %3s agpr %s sgpr %3s scratch %s" % (demangled.replace("(anonymous namespace)::", "")[:110], r.get("vgpr_count"),
                                                                    r.get("agpr_count"), r.get("sgpr_count"), r.get("private_segment_fixed_size")))
    print("%d kernels in the code object, library %.1f MB" % (total, os.path.getsize(SO) / 1e6))


if __name__ == "__main__":
    main()

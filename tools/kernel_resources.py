"""Per-kernel resource usage of the shipped code objects: VGPRs, SGPRs, LDS, scratch (private segment) and spill counts.

Reads the `.hip_fatbin` section of libdvits_hip.so, splits the offload bundle into its gfx950 ELF code objects and parses
the AMDGPU metadata note (`amdhsa.kernels`) that `llvm-readelf --notes` prints.  Used by tests/test_kernel_resources.py
(the default schedule's kernels must not use scratch memory) and as a command-line report:

    python tools/kernel_resources.py [path/to/lib.so] [name-filter]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM_BIN = os.environ.get("DVITS_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
DEFAULT_LIB = os.path.join(ROOT, "diff-vits_amd", "libdvits_hip.so")

_FIELDS = {
    ".name": str, ".symbol": str, ".vgpr_count": int, ".agpr_count": int, ".sgpr_count": int,
    ".private_segment_fixed_size": int, ".group_segment_fixed_size": int, ".vgpr_spill_count": int,
    ".sgpr_spill_count": int, ".max_flat_workgroup_size": int, ".kernarg_segment_size": int,
}


def _code_objects(lib_path, tmpdir):
    fat = os.path.join(tmpdir, "fat.bin")
    subprocess.check_call([os.path.join(LLVM_BIN, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib_path, fat])
    data = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(b"\x7fELF\x02\x01\x01", data)]
    out = []
    for k, s in enumerate(starts):
        e = starts[k + 1] if k + 1 < len(starts) else len(data)
        path = os.path.join(tmpdir, "co%d.elf" % k)
        with open(path, "wb") as f:
            f.write(data[s:e])
        out.append(path)
    return out


def _demangle(names):
    if not names:
        return {}
    import shutil
    tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    if not tool:
        return {}
    p = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True, check=True)
    return dict(zip(names, p.stdout.splitlines()))


def kernel_resources(lib_path=DEFAULT_LIB):
    """-> list of dicts (one per kernel): name (demangled), symbol, vgpr_count, sgpr_count, private_segment_fixed_size, ..."""
    kernels = []
    with tempfile.TemporaryDirectory() as tmp:
        for co in _code_objects(lib_path, tmp):
            p = subprocess.run([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", co], capture_output=True, text=True)
            if p.returncode != 0:
                continue
            cur = None
            for line in p.stdout.splitlines():
                s = line.strip()
                if s.startswith("- .agpr_count") or (s.startswith("- .") and cur is not None and ".args" not in s and _is_kernel_start(s)):
                    cur = {}
                    kernels.append(cur)
                    s = s[2:]
                elif s.startswith("- ") and cur is not None and not s.startswith("- ."):
                    continue
                if cur is None:
                    continue
                m = re.match(r"(\.[a-z_]+):\s*(.*)$", s)
                if m and m.group(1) in _FIELDS:
                    v = m.group(2).strip().strip("'\"")
                    try:
                        cur[m.group(1)[1:]] = _FIELDS[m.group(1)](v)
                    except ValueError:
                        pass
    kernels = [k for k in kernels if "name" in k and "private_segment_fixed_size" in k]
    dm = _demangle([k["name"] for k in kernels])
    for k in kernels:
        k["symbol"] = k["name"]
        k["name"] = dm.get(k["name"], k["name"])
    return kernels


def _is_kernel_start(s):
    # a kernel's map in amdhsa.kernels starts with its first key in sorted order (".agpr_count" on gfx9 code objects)
    return s.startswith("- .agpr_count")


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else DEFAULT_LIB
    flt = sys.argv[-1] if len(sys.argv) > 1 and not os.path.exists(sys.argv[-1]) else ""
    ks = [k for k in kernel_resources(lib) if flt in k["name"]]
    ks.sort(key=lambda k: (-k["private_segment_fixed_size"], k["name"]))
    print("%-6s %-6s %-6s %-8s %-8s %s" % ("vgpr", "agpr", "sgpr", "lds", "scratch", "kernel"))
    for k in ks:
        print("%-6d %-6d %-6d %-8d %-8d %s" % (k.get("vgpr_count", -1), k.get("agpr_count", -1), k.get("sgpr_count", -1),
                                              k.get("group_segment_fixed_size", -1), k["private_segment_fixed_size"], k["name"][:150]))
    print("%d kernels, %d with scratch" % (len(ks), sum(1 for k in ks if k["private_segment_fixed_size"] > 0)))


if __name__ == "__main__":
    main()

"""Cross-check: the kernels inside the shipped dm_nerf_amd/libdmnerf_hip.so are, instruction for instruction, the ISA
listings that scripts/check_asm_hazard.py examined (dm_nerf_amd/csrc/build/*-hip-amdgcn-amd-amdhsa-gfx950.s).

The device code objects are pulled out of the library with `llvm-objdump --offloading`, disassembled, and the mnemonic
sequence of every kernel is compared with the one in the compiler's listing (trailing alignment padding after the
last s_endpgm ignored).  It also reads the kernel metadata of the shipped code objects and fails if ANY kernel uses
scratch memory or spilled a register.  Usage: python scripts/verify_shipped_isa.py   (needs /opt/rocm/lib/llvm/bin; exit 1 on a
difference)."""
import collections
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")
norm = lambda m: m.replace("_e32", "").replace("_e64", "")


def from_objdump(co):
    out = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout
    k, d = None, collections.OrderedDict()
    for l in out.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", l)
        if m:
            k = m.group(1)
            d[k] = []
            continue
        m = re.match(r"^\s+([a-z][a-z0-9_]+)\b", l)
        if m and k:
            d[k].append(norm(m.group(1)))
    for v in d.values():                                 # alignment padding behind the kernel's last instruction (a kernel
        while v and v[-1] in ("s_nop", "s_code_end", "v_cndmask_b32"):   # ends in s_endpgm or a branch; zero words decode
            v.pop()                                                      # as v_cndmask_b32)
    return d


def from_listing(path):
    k, d = None, {}
    for raw in open(path):
        l = raw.split(";")[0].rstrip()
        m = re.match(r"^([A-Za-z_][\w$.]*):", l)
        if m and not m.group(1).startswith(".L"):
            k = m.group(1)
            d[k] = []
            continue
        if l.startswith(".Lfunc_end"):
            k = None
        m = re.match(r"^\s+([a-z][a-z0-9_]+)\b", l)
        if m and k and not l.strip().startswith("."):
            d[k].append(norm(m.group(1)))
    return d


def scratch_report(co):
    """Kernel metadata of one code object (llvm-readelf --notes): [(name, private_segment_fixed_size, sgpr spills, vgpr spills)]."""
    out = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
    rows, cur = [], {}
    for line in out.split("\n"):
        m = re.match(r"^\s+(?:- )?\.(name|private_segment_fixed_size|sgpr_spill_count|vgpr_spill_count):\s+(\S+)", line)
        if not m:
            continue
        cur[m.group(1)] = m.group(2)
        if len(cur) == 4:
            rows.append((cur["name"], int(cur["private_segment_fixed_size"]), int(cur["sgpr_spill_count"]), int(cur["vgpr_spill_count"])))
            cur = {}
    return rows


def main():
    tmp = tempfile.mkdtemp()
    try:
        lib = os.path.join(tmp, "lib.so")
        shutil.copy(os.path.join(ROOT, "dm_nerf_amd", "libdmnerf_hip.so"), lib)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", lib], capture_output=True, cwd=tmp, check=True)
        shipped = {}
        spilled = []
        n_meta = 0
        for co in sorted(glob.glob(lib + ".*gfx950")):
            shipped.update(from_objdump(co))
            for name, scratch, s_sp, v_sp in scratch_report(co):
                n_meta += 1
                if scratch or s_sp or v_sp:
                    spilled.append((name, scratch, s_sp, v_sp))
                    print(f"SCRATCH/SPILL in the shipped library: {name}: {scratch} B scratch, {v_sp} VGPR / {s_sp} SGPR spills")
        same = diff = 0
        for s in sorted(glob.glob(os.path.join(ROOT, "dm_nerf_amd", "csrc", "build", "*-hip-amdgcn-amd-amdhsa-gfx950.s"))):
            for k, v in from_listing(s).items():
                if k not in shipped:
                    continue
                ok = v == shipped[k]
                same += ok
                diff += not ok
                if not ok:
                    print("DIFFERENT", k, len(v), len(shipped[k]))
        print(f"{same} kernels identical to their checked listing, {diff} different")
        print(f"{n_meta} kernels in the shipped code objects, {len(spilled)} with scratch memory or register spills")
        return 1 if diff or not same or spilled or not n_meta else 0
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())

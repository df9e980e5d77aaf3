"""Build gate: no shipped kernel may use scratch memory or spill a register.

Reads the AMDGPU metadata the compiler writes at the end of each ISA listing (``-save-temps``) and fails when any kernel
has ``.private_segment_fixed_size``, ``.vgpr_spill_count`` or ``.sgpr_spill_count`` different from 0.  The fused MLP
kernels run one wave per SIMD with 448-500 of the 512 registers: a spill there is a silent round trip through HBM-backed
scratch in the middle of an MFMA stream (round 1 shipped 30 spilled VGPRs in the training forward and 37 / 67 spilled
SGPRs in the Replica-width dgrad kernels; both were artefacts of hoisted address arithmetic, see mlp_fwd_impl.h ``fresh``).

Usage: python scripts/check_no_scratch.py file.s [...]      (exit 1 on a violation; used by dm_nerf_amd/csrc/Makefile)"""
import re
import sys


def kernels(path):
    """-> {kernel name: {field: int}} from the ``amdhsa.kernels`` metadata of one listing."""
    out, cur = {}, {}
    for line in open(path):
        m = re.match(r"^\s+(?:- )?\.(name|private_segment_fixed_size|sgpr_spill_count|vgpr_spill_count|sgpr_count|vgpr_count|agpr_count):\s+(\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "name":
            cur["name"] = v
        else:
            cur[k] = int(v)
        if "name" in cur and all(f in cur for f in ("private_segment_fixed_size", "sgpr_spill_count", "vgpr_spill_count")):
            out[cur.pop("name")] = cur
            cur = {}
    return out


def violations(path):
    return [(name, f) for name, f in kernels(path).items()
            if f["private_segment_fixed_size"] or f["sgpr_spill_count"] or f["vgpr_spill_count"]]


def main(argv):
    bad = 0
    n = 0
    for path in argv:
        ks = kernels(path)
        n += len(ks)
        for name, f in violations(path):
            bad += 1
            print(f"SCRATCH/SPILL {path}: {name}: scratch {f['private_segment_fixed_size']} B, "
                  f"{f['vgpr_spill_count']} VGPR / {f['sgpr_spill_count']} SGPR spills")
    print(f"{n} kernel(s), {bad} with scratch or spills in {len(argv)} file(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))

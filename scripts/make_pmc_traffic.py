"""profiles/pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of scripts/gpu_profile.sh.

Usage: python scripts/make_pmc_traffic.py gpurun_out/prof_<tag> <source note>
Per the MI355X guide: separate --pmc passes, FETCH_SIZE/WRITE_SIZE in KB, FETCH_SIZE x2 on gfx950 for
16 B/lane streams.  Per launch of the dominant kernel (fine-network inference MLP, grid 1 572 864 threads).
"""
import csv, glob, json, os, sys

out, note = sys.argv[1], sys.argv[2]
KERNEL, GRID = "mlp_fwd_kernel<1, false, false, false>", 1572864


def mean(sub, counter):
    p = glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(p))
         if r["Counter_Name"] == counter and KERNEL in r["Kernel_Name"] and int(r["Grid_Size"]) == GRID]
    return sum(v) / len(v), len(v)


f, nf = mean("pmc_fetch", "FETCH_SIZE")
w, nw = mean("pmc_write", "WRITE_SIZE")
d = {"round": 6, "kernel": "mlp_fwd_kernel<1,false,false,false> fine launch (4096x192 samples)", "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
     "dispatches": [nf, nw], "fetch_correction": 2.0, "traffic_bytes_per_launch": int(round((2.0 * f + w) * 1024)), "source": note}
json.dump(d, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json"), "w"), indent=1)
print(d)

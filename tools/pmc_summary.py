"""tools/pmc_summary.py -- condense a rocprofv3 output dir (kernel stats / counter collection CSVs) to a
small JSON: per knhip kernel, dispatch count and mean counter value per dispatch."""
import collections, csv, glob, json, sys
src, dst = sys.argv[1], sys.argv[2]
out = {}
for f in glob.glob(f"{src}/**/*_counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "knhip" not in kn:
            continue
        agg[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[kn] = dict(grid=r["Grid_Size"], wg=r["Workgroup_Size"], lds=r["LDS_Block_Size"], vgpr=r["VGPR_Count"],
                        sgpr=r["SGPR_Count"], scratch=r["Scratch_Size"])
    for kn, cs in agg.items():
        out.setdefault(kn, {"meta": meta[kn], "counters": {}})
        for c, v in cs.items():
            out[kn]["counters"][c] = {"dispatches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
for f in glob.glob(f"{src}/**/*_kernel_stats.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "knhip" in r["Name"]]
    out["__kernel_stats__"] = rows
json.dump(out, open(dst, "w"), indent=1)
print("wrote", dst, len(out), "kernels")

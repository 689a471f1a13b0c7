"""tools/all_kernels_md.py TAG STEPS OUT.md -- one table of every kernel of the profiled bench run (gpurun_out/TAG_*.json written by
tools/profile_bench.sh): calls and time per step from --kernel-trace --stats, and per launch the counters of the separate
--pmc passes (HBM bytes with the gfx950 corrections, matrix-pipe busy, LDS active / bank conflicts, waves waiting)."""
import json, sys, os
tag, steps, out = sys.argv[1], float(sys.argv[2]), sys.argv[3]
def load(n):
    p = f"gpurun_out/{tag}_{n}.json"
    return json.load(open(p)) if os.path.exists(p) else {}
stats = load("stats").get("__kernel_stats__", [])
pm = {n: load(n) for n in ("fetch", "write", "sq", "lds", "mfma", "wait")}
def ctr(n, k, c):
    return pm[n].get(k, {}).get("counters", {}).get(c, {}).get("mean")
rows = []
for r in stats:
    k = r["Name"]
    calls, tot = int(r["Calls"]), float(r["TotalDurationNs"])
    if tot / steps < 2000:  # (under 2 us per step: index build leftovers and one-off kernels are listed by the stats file)
        continue
    f, w = ctr("fetch", k, "FETCH_SIZE"), ctr("write", k, "WRITE_SIZE")
    gb = ((f or 0) * 1024 * 2 + (w or 0) * 1024) / 1e9 if (f is not None or w is not None) else None
    gui, mb = ctr("mfma", k, "GRBM_GUI_ACTIVE"), ctr("mfma", k, "SQ_VALU_MFMA_BUSY_CYCLES")
    mf = mb / (gui / 8 * 1024) if gui and mb is not None else None  # (GRBM cycles summed over 8 XCDs; 1024 SIMDs)
    wc, wi = ctr("sq", k, "SQ_WAVE_CYCLES"), ctr("sq", k, "SQ_WAIT_INST_ANY")
    la, lc = ctr("lds", k, "SQ_LDS_IDX_ACTIVE"), ctr("lds", k, "SQ_LDS_BANK_CONFLICT")
    rows.append((tot / steps / 1e6, k.split("(")[0].replace("void ", "")[:70], calls / steps, tot / calls / 1e3, gb, mf,
                 wi / wc if wc and wi is not None else None, lc / la if la and lc is not None else None))
rows.sort(reverse=True)
# the profiled run BUILDS the index first (k-means, assignment of every base row through the coarse stage, encode): kernels
# launched more than a few times per step belong to it (the coarse kernels run in both; their search-time cost is the
# `coarse` stage of the bench line)
build = [r for r in rows if r[2] > 6]
rows = [r for r in rows if r[2] <= 6]
fmt = lambda v, p: "" if v is None else (p % v)
with open(out, "w") as o:
    o.write(f"# every kernel of the profiled bench run `{tag}` (per step; kernels under 2 us per step left out)\n\n")
    o.write("| ms / step | kernel | launches / step | us / launch | HBM GB / launch (PMC) | matrix pipe busy | waves waiting | LDS cycles lost to bank conflicts |\n|---|---|---|---|---|---|---|---|\n")
    for ms, k, n, us, gb, mf, wt, lc in rows:
        o.write(f"| {ms:.3f} | `{k}` | {n:.1f} | {us:.1f} | {fmt(gb, '%.3f')} | {fmt(mf, '%.2f')} | {fmt(wt, '%.2f')} | {fmt(lc, '%.2f')} |\n")
    o.write(f"\nsum {sum(r[0] for r in rows):.3f} ms per step under the profiler, without the coarse stage (below).  Launches per step of 1.2 = 4 timed + warm-up steps and the recall check; 0.2 = once (layouts built on first use).\n")
    o.write("\n## kernels of the index build in the same run (per launch; the coarse stage's kernels also run once per search step)\n\n")
    o.write("| kernel | launches in the run | us / launch (build-dominated average) | HBM GB / launch | matrix pipe busy | waves waiting |\n|---|---|---|---|---|---|\n")
    for ms, k, n, us, gb, mf, wt, lc in build:
        o.write(f"| `{k}` | {n * steps:.0f} | {us:.1f} | {fmt(gb, '%.3f')} | {fmt(mf, '%.2f')} | {fmt(wt, '%.2f')} |\n")
print(open(out).read()[:3000])

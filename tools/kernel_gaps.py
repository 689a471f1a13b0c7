"""tools/kernel_gaps.py <rocprofv3 output dir> <anchor kernel substring> <out.txt> -- the timeline of ONE step from a
rocprofv3 --kernel-trace run: every dispatch between two consecutive launches of the anchor kernel (the step's dominant
kernel), with its duration and the idle gap in front of it.  Shows where a step loses time BETWEEN kernels (host
round trips, launch-bound stretches), which per-kernel statistics cannot."""
import csv, glob, os, sys
d, anchor, out = sys.argv[1:4]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
idx = [i for i, r in enumerate(rows) if anchor in r[2] and (r[1] - r[0]) > 1_000_000]
with open(out, "w") as o:
    if len(idx) < 4:
        o.write(f"fewer than 4 long launches of {anchor}\n")
        sys.exit(0)
    a, b = idx[-3], idx[-2]  # (a late step: warm)
    t0 = rows[a][0]
    o.write(f"one step = dispatches [{a}, {b}): {(rows[b][0] - t0) / 1e3:.1f} us from anchor start to anchor start\n")
    busy = gap_tot = 0
    prev_end = rows[a][0]
    for s, e, n in rows[a:b]:
        gap = s - prev_end
        o.write(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {gap / 1e3:7.1f}  {n[:90]}\n")
        busy += e - s
        gap_tot += max(gap, 0)
        prev_end = max(prev_end, e)
    o.write(f"busy {busy / 1e3:.1f} us, idle gaps {gap_tot / 1e3:.1f} us (side-stream kernels overlap: negative gaps)\n")

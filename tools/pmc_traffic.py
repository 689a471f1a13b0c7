"""tools/pmc_traffic.py -- the PMC evidence of the dominant scan kernel from the passes of tools/profile_bench.sh ->
profiles/bench_pmc_traffic.json (key = the bench workload [+ kernel path], value stamped with the commit): HBM bytes per
launch from FETCH_SIZE / WRITE_SIZE, and -- when the sq / lds / mfma summaries are given -- the issue counters bench.py
turns into unit-busy fractions (SQ_LDS_IDX_ACTIVE, SQ_INSTS_VALU, SQ_VALU_MFMA_BUSY_CYCLES against GRBM_GUI_ACTIVE / 8).
usage: python tools/pmc_traffic.py <fetch.json> <write.json> <key> <kernel-substring> [commit] [sq.json lds.json mfma.json]"""
import json, os, subprocess, sys
fetch, write, key, kern = sys.argv[1:5]
commit = sys.argv[5] if len(sys.argv) > 5 and sys.argv[5] else subprocess.run(
    ["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
extra = sys.argv[6:]


def mean(path, counter):
    d = json.load(open(path))
    best = None
    for k, v in d.items():
        if kern in k and counter in v.get("counters", {}):
            c = v["counters"][counter]
            # (max over the dispatches: the prefilter kernels are launched a second time per step for the -- normally
            # empty -- retry round; the mean would halve the main launch's traffic)
            if best is None or c["max"] > best[0]:
                best = (c["max"], k, c["dispatches"])
    return best


f, w = mean(fetch, "FETCH_SIZE"), mean(write, "WRITE_SIZE")
# gfx950: FETCH_SIZE (KB) reports half of a wide coalesced read (MI355X_MICROARCH.md, HBM section); WRITE_SIZE uncalibrated
bytes_ = f[0] * 1024 * 2 + (w[0] * 1024 if w else 0)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(root, "profiles", "bench_pmc_traffic.json")
try:
    t = json.load(open(path))
except Exception:
    t = {}
t[key] = {"hbm_bytes_per_launch": bytes_, "kernel": f[1], "fetch_size_kb": f[0], "write_size_kb": w[0] if w else None,
          "dispatches": f[2], "commit": commit, "correction": "FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024 (largest dispatch)"}
if extra:
    sq = {}
    for p in extra:
        d = json.load(open(p))
        for k, v in d.items():
            if k == f[1]:
                for c, cv in v.get("counters", {}).items():
                    # the LARGEST dispatch of every counter, as for the traffic: the prefilter kernels have an (empty)
                    # second launch per step, a mean over the dispatches halves every absolute count
                    sq[c] = cv.get("max", cv["mean"])
                    sq.setdefault("_dispatch_stat", "max")
    sq["source"] = ", ".join(os.path.basename(p) for p in extra) + f" (separate --pmc passes of the bench, commit {commit})"
    t[key]["sq"] = sq
json.dump(t, open(path, "w"), indent=1)
print(key, bytes_)

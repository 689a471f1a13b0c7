"""tools/check_pqd_ring.py <pq_decode.s> -- build-time check of pq_decode.hip's hand-managed registers.

The tile loop keeps its ring of loads in NAMED accumulation registers a[216:255] that only inline ISA touches (the compiler
cannot follow loads that turn by name through an unrolled loop; pq_decode.hip, "the ring loads").  Every ring statement
lists them as clobbered, which keeps the compiler from holding a value in them ACROSS such a statement -- but nothing keeps
it from using them for a value that lives BETWEEN two statements, which would silently destroy a slot.  It has no reason to
while it needs fewer than 216 accumulation registers; this script makes the build fail the day it does: no instruction
outside the inline-ISA blocks of a pqd_kernel may name a216 .. a255."""
import re
import sys

RING_LO = 216
src = open(sys.argv[1]).read().split("\n")
fn, inasm, bad, hi = None, False, [], {}
for n, l in enumerate(src, 1):
    m = re.match(r"^(_ZN5knhip10pqd_kernel\w+):", l)
    if m:
        fn = m.group(1)
        continue
    if fn and l.startswith(".Lfunc_end"):
        fn = None
        continue
    if not fn:
        continue
    if "ASMSTART" in l:
        inasm = True
        continue
    if "ASMEND" in l:
        inasm = False
        continue
    if inasm:
        continue
    code = l.split(";")[0]
    for m in re.finditer(r"\ba(\d+)\b|\ba\[(\d+):(\d+)\]", code):
        top = int(m.group(1)) if m.group(1) else int(m.group(3))
        hi[fn] = max(hi.get(fn, -1), top)
        if top >= RING_LO:
            bad.append((fn, n, l.strip()))
if not hi:
    sys.exit("check_pqd_ring: no pqd_kernel found in " + sys.argv[1])
for f, h in sorted(hi.items()):
    print(f"check_pqd_ring: {f}: compiler-managed accumulation registers up to a{h} (ring: a{RING_LO}..a255)")
if bad:
    for f, n, l in bad[:10]:
        print(f"check_pqd_ring: {f}: line {n}: {l}", file=sys.stderr)
    sys.exit("check_pqd_ring: the compiler uses the ring's named registers -- move the ring or lower the register pressure")

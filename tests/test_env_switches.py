"""The library reads its KNHIP_* switches in one place (knowhere_amd/csrc/knhip_env.h); DESIGN 4.6 documents each of them."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "knowhere_amd", "csrc")


def test_only_the_env_header_reads_the_environment():
    offenders = []
    for path in glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")):
        if os.path.basename(path) == "knhip_env.h":
            continue
        for n, line in enumerate(open(path), 1):
            if re.search(r"\bgetenv\s*\(", line):
                offenders.append(f"{os.path.basename(path)}:{n}")
    assert not offenders, offenders


def test_every_switch_of_the_header_is_documented():
    names = set(re.findall(r'getenv\("(KNHIP_[A-Z0-9_]+)"\)', open(os.path.join(CSRC, "knhip_env.h")).read()))
    assert len(names) >= 20
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    missing = sorted(n for n in names if n not in design)
    assert not missing, missing

#!/usr/bin/env python3
"""Assembles DESIGN.md (the current-state document) from tools/design/DESIGN.in.md, carrying over verbatim the sections of
HISTORY.md that describe stable parts of the design (the path and its boundary, the floating-point specification, the
multi-GPU exchange, the oracle's status): those are not round narrative, and a second hand-maintained copy would drift.
Run after editing a part:  python tools/make_design.py"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = open(os.path.join(ROOT, "HISTORY.md")).read()


def section(start, end):
    a = H.index(start)
    return H[a:H.index(end, a)].rstrip() + "\n"


CARRIED = {
    "@@SECTION_1@@": ("## 1. The path and its boundary (SURVEY.md §8 a, b)", "## 2. Data layout in HBM"),
    "@@SECTION_2@@": ("## 2. Data layout in HBM", "## 3. Kernel 1"),
    "@@SECTION_3_2@@": ("### 3.2 Floating-point specification", "### 3.3 Resources and bound"),
    "@@SECTION_7@@": ("## 7. Multi-GPU (SURVEY.md §8 e)", "## 8. Oracle and parity status"),
    "@@SECTION_8@@": ("## 8. Oracle and parity status (SURVEY.md §8 c)", "## 9. Out of scope"),
}
text = open(os.path.join(ROOT, "tools", "design", "DESIGN.in.md")).read()
for k, (a, b) in CARRIED.items():
    # cross references of a carried section to sub-sections that now live in HISTORY.md only
    body = re.sub(r"§(3\.3|3\.4|4\.3|4\.4|5\.[0-9]+|6\.[0-9]|13\.1|13\.2|14\.1)\b", lambda m: "HISTORY §" + m.group(1), section(a, b))
    text = text.replace(k, body)
open(os.path.join(ROOT, "DESIGN.md"), "w").write(text)
print("DESIGN.md:", len(text.splitlines()), "lines")

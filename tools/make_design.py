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
# statements of the carried sections that later rounds overtook (the history keeps them as they were written)
for old, new in [
    ("ONE banded solve of the KKT system with blocks of nq + nu (`csrc/kkt.h`, HISTORY §13.1; blocks ≤ 24); the reference's route over S = J H⁻¹ Jᵀ (`csrc/constraints.h`, `dense_ldl.h`) for allegro and behind the stepwise API",
     "ONE banded solve of the KKT system with blocks of nq + nu (`csrc/kkt.h`, §5.5, HISTORY §13.1; blocks ≤ 30: every example, allegro's 23 + 6 included); the reference's route over S = J H⁻¹ Jᵀ (`csrc/constraints.h`, `dense_ldl.h`) behind the stepwise API and for a singular S"),
    ("solve with H run on the device.  Without convergence checks and with the non-adaptive scalings\n  — all five example configurations — the whole trust-region loop incl. the multipliers of enforced\n  equality constraints runs on the device and the host waits once per `Solve` (§13); with\n  convergence checks, the adaptive scalings or `IDTO_OPT_HOST_LOOP=1` the O(num_vars) bookkeeping",
     "solve with H run on the device.  The whole trust-region loop - the multipliers of enforced equality\n  constraints, the convergence criteria and (with diagonal cost weights) the adaptive scalings included - runs on the\n  device and the host waits once per `Solve` (§13: all five example configurations); with dense cost weights under\n  an adaptive scaling, `linear_solver = kDenseLdlt`, the debug switches or `IDTO_OPT_HOST_LOOP=1` the O(num_vars) bookkeeping"),
    ("`tr_iter_kernel`, `tr_decide` in `cost_kernel` (`csrc/trust_region.h`, §13)",
     "`tr_iter_kernel`, `tr_decide` in `cost_kernel` (`csrc/trust_region.h`, §13); acrobot and the spinner: the whole iteration in `gn_small_kernel` (§5.4)"),
]:
    if old in text:
        text = text.replace(old, new)
open(os.path.join(ROOT, "DESIGN.md"), "w").write(text)
print("DESIGN.md:", len(text.splitlines()), "lines")

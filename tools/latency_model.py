#!/usr/bin/env python3
"""The bounds that actually apply to the two kernels of a Gauss-Newton step - they sit at 0.3 % of the HBM roofline, so
the HBM fraction of bench.py's `roofline` says nothing actionable - computed from the committed profiles
of a round and written to profiles/<round>_latency_model.json, which bench.py reads (the newest one) and reports next to
the kernel times it measures live.

  fd_kernel          VALU issue floor: instructions per wavefront (SQ_INSTS_VALU / SQ_WAVES of <round>_pmc_sq_summary.txt)
                     x 4 cycles (one wavefront per SIMD; an f64 or a 32-bit VALU instruction occupies the SIMD for 4 cycles:
                     SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.0 quad-cycles in the same summary) at 2.4 GHz
  penta_pipe_kernel  (a) the dependent pivot chain: (block rows of the longest chain + the separator's 2) x K pivots x the
                     measured 80-cycle link (profiles/r03_microbench.txt, tools/micro/chain_bench.hip);
                     (b) the row model: rows x measured row-to-row time + join + separator + back substitution, all from
                     the in-kernel stamps of <round>_nd_timeline.txt - what the launch takes if nothing but the rows'
                     own work is removed

usage: python tools/latency_model.py r04"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLOCK_MHZ = 2400.0
PIVOT_LINK_CYCLES = 80   # fallback; the round's own <round>_microbench.txt ("pivot link of penta_pipe.h ...": tools/micro/chain_bench.hip) is read when it is there


def pmc(path, kernel):
    out = {}
    for line in open(path):
        m = re.match(r"(pmc_sq\d) (\S+) (\{.*\})", line.strip())
        if m and m.group(2) == kernel:
            out.update(eval(m.group(3)))
    return out


def kernel_avg_us(path, kernel):
    for row in csv.DictReader(open(path)):
        if kernel in row["Name"]:
            return float(row["AverageNs"]) / 1e3
    return None


def timeline(path):
    chains, sep = [], None
    cur = None
    for line in open(path):
        m = re.match(r"(P0|P3|J1|J2) (producer|joiner)\s+start\s+([-\d.]+).*forward done\s+([-\d.]+)\s+backward start\s+([-\d.]+)\s+end\s+([-\d.]+)", line)
        if m:
            cur = {"name": m.group(1), "role": m.group(2), "start": float(m.group(3)), "forward_done": float(m.group(4)),
                   "backward_start": float(m.group(5)), "end": float(m.group(6))}
            chains.append(cur)
            continue
        m = re.search(r"elimination of row il \(start, end\):(.*)", line)
        if m and cur is not None:
            cur["rows"] = [(float(a), float(b)) for a, b in re.findall(r"\(\s*([-\d.]+),\s*([-\d.]+)\)", m.group(1))]
        m = re.search(r"median: row to row ([\d.]+) us, the K pivots ([\d.]+) us", line)
        if m and cur is not None:
            cur["row_to_row_us"], cur["k_pivots_us"] = float(m.group(1)), float(m.group(2))
        m = re.search(r"corrected by the separator's solution ([-\d.]+), recursion from ([-\d.]+) to ([-\d.]+)", line)
        if m and cur is not None:
            cur["corrected"], cur["recursion_from"], cur["recursion_to"] = (float(m.group(i)) for i in (1, 2, 3))
        m = re.match(r"separator\s+start\s+([-\d.]+)\s+Q ready\s+([-\d.]+).*solved\+posted\s+([-\d.]+)", line)
        if m:
            sep = {"q_ready": float(m.group(2)), "solved": float(m.group(3))}
    return chains, sep


def pipe8_prediction(chains, sep, n_rows=40, K=19):
    """cost a third dissection level - eight chains of ~4-5 rows instead of four of ~10 - BEFORE building it,
    from the terms this round's timeline measures (solver alone, no assembly in front).  Build only if <= 44 us.

    Terms (all from <round>_nd_timeline.txt): t0 first pivot of a chain; r_P / r_J row-to-row time of a chain without /
    with spike columns; pseudo = the spike-less chain's pseudo-rows and their stores; join_wait, join_rows = the pair's
    meeting (contributions in LDS; the two join rows); hand = last join row -> Q summed at the separator; sep = the
    separator's two rows + solve + post; hop = separator's answer -> a chain corrected; rec = recursion per row; phop =
    join rows of x -> the pair's other chain.

    Two ways to get eight chains:
      flat    three separators s1 s2 s3 at the quarter points, four twisted pairs; the reduced system [s1 s2 s3] is block
              tridiagonal in 2K x 2K blocks: s1 and s3 eliminated side by side, their Schur complements handed to s2, s2
              solved, s1 / s3 substituted, then the chains.
      nested  today's separator kept, each half gets a sub-separator of its own (two twisted pairs per half); the
              sub-separator's two rows carry 2K fill columns to the main separator.
    In both, every chain next to a separator carries spike columns (r_J, not r_P), and the join rows of a segment between
    TWO separators carry 4K spike columns: their elimination is taken at (1 + 2K / (3K + 1)) x today's join row - the
    spike wavefronts' share doubles; generous, since 4K columns do not fit the three spike wavefronts' lanes (2 x 38 > 64)
    nor the LDS carve-up (163.2 of 163.8 KB at K = 19 today)."""
    P = next(c for c in chains if c["role"] == "producer")
    J = max((c for c in chains if c["role"] == "joiner"), key=lambda c: c["forward_done"])
    t0 = J["rows"][0][0]
    r_P, r_J = P["row_to_row_us"], J["row_to_row_us"]
    pseudo = P["forward_done"] - P["rows"][-1][1]
    pre = len(J["rows"]) - 2
    join_start = J["rows"][pre][0]                    # first join row's elimination begins
    join_wait = join_start - J["rows"][pre - 1][1]    # last pre-join row done -> first join row under way
    join_rows = J["forward_done"] - join_start
    hand = sep["q_ready"] - J["forward_done"]
    sep_us = sep["solved"] - sep["q_ready"]
    hop = J["corrected"] - sep["solved"]
    rec = (J["recursion_to"] - J["recursion_from"]) / len(J["rows"])
    phop = P["recursion_from"] - (J["recursion_from"] + 2 * rec)
    wide = 1.0 + 2.0 * K / (3.0 * K + 1.0)
    today = t0 + pre * r_J + join_wait + join_rows + hand + sep_us + hop + (J["recursion_from"] - J["corrected"]) + \
        2 * rec + phop + len(P["rows"]) * rec
    # ---- flat: 40 - 6 = 34 rows in segments of 8, 9, 9, 8; an inner segment (9 rows): 3 + 4 pre-join rows, 2 join rows
    outer = t0 + max(3 * r_P + pseudo, 3 * r_J) + join_wait + join_rows
    inner = t0 + 4 * r_J + join_wait + wide * join_rows
    elim = sep_us - 1.5                                # a separator's two rows without its solve + post (timeline: 1.5 us)
    q13 = max(outer, inner) + hand                     # s1 / s3 have their Q
    s2_in = q13 + elim + hand                          # their Schur complements summed at s2
    x2 = s2_in + sep_us
    x13 = x2 + hop + 1.5                               # s1 / s3: one mat-vec, two triangular solves, post
    flat = x13 + hop + 2 * rec + phop + 4 * rec
    # ---- nested: a half = 19 rows = sub-separator (2) + segments of 8 (outer) and 9 (between the two separators)
    sub_q = max(outer, inner) + hand
    sub_done = sub_q + elim * wide                     # its two rows carry 2K more columns
    main_q = sub_done + hand
    xm = main_q + sep_us
    xs = xm + hop + 1.5
    nested = xs + hop + 2 * rec + phop + 4 * rec
    return {
        "what": "eight chains instead of four (a third dissection level), predicted from this round's measured terms; solver alone",
        "terms_us": {"first_pivot": t0, "row_without_spikes": r_P, "row_with_spikes": r_J, "pseudo_rows": pseudo, "join_wait": join_wait,
                     "two_join_rows": join_rows, "hand_over_to_separator": hand, "separator": sep_us, "answer_to_chain": hop,
                     "recursion_per_row": rec, "join_rows_of_x_to_the_other_chain": phop, "join_row_factor_with_4K_spike_columns": wide},
        "model_of_todays_kernel_us": today, "timeline_of_todays_kernel_us": max(c["end"] for c in chains),
        "flat_three_separators_us": flat, "nested_sub_separators_us": nested,
        "segments_forward_done_us": {"one_separator": outer, "two_separators": inner},
        "threshold_us": 44.0, "build": bool(min(flat, nested) <= 44.0),
        "why": "halving the chains saves ~6 rows (16-20 us) and buys a second reduced-system level: one more hand-over + "
               "separator elimination + hand-over going up, one more answer + substitution going down, and join rows with "
               "twice the spike columns in every segment that sits between two separators - within 2-3 us of what the rows save",
    }


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
    P = lambda f: os.path.join(ROOT, "profiles", f"{rnd}_{f}")
    out = {"round": rnd, "clock_mhz": CLOCK_MHZ, "sources": {}}
    # ---- fd_kernel
    c = pmc(P("pmc_sq_summary.txt"), "fd_kernel")
    valu = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
    fd_us = kernel_avg_us(P("kernel_stats.csv"), "fd_kernel")
    floor = valu * 4 / CLOCK_MHZ
    out["fd_kernel"] = {
        "bound": "VALU issue, one wavefront per SIMD (4 cycles per instruction)",
        "valu_instructions_per_wavefront": round(valu), "issue_floor_us": floor, "rocprof_avg_us": fd_us,
        "achieved_frac_of_issue_floor": floor / fd_us if fd_us else None,
        "wave_cycles_per_wavefront": 4 * c["SQ_WAVE_CYCLES"] / c["SQ_WAVES"],
        "wait_frac_of_wave_cycles": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
        "valu_active_frac_of_wave_cycles": c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"],
    }
    out["sources"]["fd_kernel"] = [f"profiles/{rnd}_pmc_sq_summary.txt", f"profiles/{rnd}_kernel_stats.csv"]
    # ---- penta_pipe_kernel
    chains, sep = timeline(P("nd_timeline.txt"))
    sol_us = kernel_avg_us(P("kernel_stats_assembly_in_its_own_launch.csv"), "penta_pipe_kernel")
    K = 19
    link, link_src = PIVOT_LINK_CYCLES, "profiles/r03_microbench.txt"
    if os.path.exists(P("microbench.txt")):   # (VERDICT r4 hygiene: the chain link from this round's microbench, not round 3's)
        for line in open(P("microbench.txt")):
            m = re.search(r"pivot link of penta_pipe\.h.*?\s([\d.]+) cycles", line)
            if m:
                link, link_src = float(m.group(1)), f"profiles/{rnd}_microbench.txt"
    longest = max(chains, key=lambda ch: ch["forward_done"])
    rows = len(longest["rows"])
    chain_floor = (rows + 2) * K * link / CLOCK_MHZ
    first_row = longest["rows"][0][0]
    row_model = first_row + rows * longest["row_to_row_us"] + (sep["q_ready"] - longest["forward_done"]) + (sep["solved"] - sep["q_ready"]) + \
        (max(ch["end"] for ch in chains) - sep["solved"])
    out["penta_pipe_kernel"] = {
        "bound": "dependent block rows of the longest chain, then the separator, then the back substitution",
        "block_size_K": K, "rows_of_the_longest_chain": rows, "chain": longest["name"],
        "row_to_row_us": longest["row_to_row_us"], "k_pivots_us": longest["k_pivots_us"],
        "hand_over_to_separator_us": sep["q_ready"] - longest["forward_done"], "separator_us": sep["solved"] - sep["q_ready"],
        "back_substitution_us": max(ch["end"] for ch in chains) - sep["solved"],
        "row_model_us": row_model, "pivot_chain_floor_us": chain_floor, "rocprof_avg_us_solver_alone": sol_us,
        "achieved_frac_of_row_model": row_model / sol_us if sol_us else None,
        "achieved_frac_of_pivot_chain_floor": chain_floor / sol_us if sol_us else None,
    }
    out["sources"]["penta_pipe_kernel"] = [f"profiles/{rnd}_nd_timeline.txt", f"profiles/{rnd}_kernel_stats_assembly_in_its_own_launch.csv",
                                           link_src]
    out["penta_pipe_kernel"]["pivot_link_cycles"] = link
    out["penta_pipe8_predicted"] = pipe8_prediction(chains, sep)
    out["sources"]["penta_pipe8_predicted"] = [f"profiles/{rnd}_nd_timeline.txt"]
    # ---- penta_nd_kernel<23> (allegro_hand N = 60: the seven-workgroup kernel of penta_nd.h), from its own timeline
    # (tools/nd_timeline.py allegro_hand 60 -> <round>_nd_timeline_allegro.txt): VERDICT r4 "weak" #3 asked where its 127 us go
    if os.path.exists(P("nd_timeline_allegro.txt")):
        txt = open(P("nd_timeline_allegro.txt")).read()
        ch = {m.group(1): {"forward_done": float(m.group(2)), "backward_start": float(m.group(3)), "end": float(m.group(4))}
              for m in re.finditer(r"(P0|P3|J1|J2) (?:producer|joiner)\s+start\s+[-\d.]+.*?forward done\s+([-\d.]+)\s+backward start\s+([-\d.]+)\s+end\s+([-\d.]+)", txt)}
        spikes = [[(float(a), float(b)) for a, b in re.findall(r"\(\s*([\d.]+),\s*([\d.]+)\)", m.group(1))]
                  for m in re.finditer(r"rows \(ready, done\):(.*)", txt)]
        pub = [float(x) for x in re.findall(r"last row published\s+([\d.]+)", txt)]
        sm = re.search(r"separator\s+start\s+[-\d.]+\s+Q ready\s+([\d.]+).*solved\+posted\s+([\d.]+)", txt)
        if ch and spikes and sm:
            jf = max(ch["J1"]["forward_done"], ch["J2"]["forward_done"])
            nrows = len(spikes[0])
            spike_row = sum(b - a for sp in spikes for a, b in sp) / sum(len(sp) for sp in spikes)
            q_ready, solved = float(sm.group(1)), float(sm.group(2))
            end = max(c["end"] for c in ch.values())
            stats = P("all_configs.txt")
            nd_us = None
            if os.path.exists(stats):
                m = re.search(r"allegro_hand N=60:.*penta_nd_kernel ([\d.]+)", open(stats).read())
                nd_us = float(m.group(1)) if m else None
            out["penta_nd_kernel_23"] = {
                "bound": "the chains' forward pass (producers: 14 rows + 2 pseudo-rows, joiners: 13 rows + 2 join rows), then the spike "
                         "workgroups' last two rows (they follow their joiner row by row), the separator's input and its two rows, "
                         "and the back substitution in recursion form (DESIGN.md 5.13)",
                "block_size_K": 23, "rows_of_a_joiner": nrows,
                "joiner_row_us": jf / nrows, "spike_row_us": spike_row, "first_spike_row_ready_us": min(sp[0][0] for sp in spikes),
                "joiners_forward_done_us": jf, "last_spike_row_published_us": max(pub),
                "joiners_done_to_q_ready_us": q_ready - jf, "separator_us": solved - q_ready,
                "back_substitution_us": end - solved,
                "timeline_end_us": end, "hip_event_avg_us": nd_us,
            }
            out["sources"]["penta_nd_kernel_23"] = [f"profiles/{rnd}_nd_timeline_allegro.txt", f"profiles/{rnd}_all_configs.txt"]
    json.dump(out, open(P("latency_model.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

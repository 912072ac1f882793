#!/usr/bin/env python3
"""Fills the @@R6_...@@ placeholders of tools/design/DESIGN.in.md from the committed profiles of the round and runs
tools/make_design.py.  The numbers in DESIGN.md are therefore the ones in profiles/: usage  python tools/fill_design.py r06"""
import csv
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = lambda f: os.path.join(ROOT, "profiles", f"{R}_{f}")


def avg_us(path, kernel):
    for row in csv.DictReader(open(path)):
        if kernel in row["Name"]:
            return float(row["AverageNs"]) / 1e3
    raise SystemExit(f"{kernel} not in {path}")


def small_iter_table():
    """profiles/<round>_small_iteration_times.txt as a table: us per iteration by launch structure"""
    t = {}
    for m in re.finditer(r"(\w+) constraints (off|enforced) tr_small=(\d) tr_fold=(\d): ([\d.]+) ms", open(P("small_iteration_times.txt")).read()):
        t.setdefault((m.group(1), m.group(2)), {})[(m.group(3), m.group(4))] = 1e3 * float(m.group(5))
    rows = ["| µs per iteration (N = 40, scaling on) | ONE launch (`tr_fold`) | two: `tr_iter_kernel` + `gn_small_kernel` | the loop of 4 / 7 launches |", "|---|---|---|---|"]
    for (name, con), v in t.items():
        rows.append(f"| {name}, constraint {con} | **{v[('1', '1')]:.1f}** | {v[('1', '0')]:.1f} | {v[('0', '0')]:.1f} |")
    return "\n" + "\n".join(rows) + "\n"


b = json.load(open(P("bench.json")))
lm = json.load(open(P("latency_model.json")))
tr = json.load(open(P("pmc_traffic.json")))
cfgs = open(P("all_configs.txt")).read()
fd_us, pipe_us = avg_us(P("kernel_stats.csv"), "fd_kernel"), avg_us(P("kernel_stats.csv"), "penta_pipe_kernel")
pipe_alone = avg_us(P("kernel_stats_assembly_in_its_own_launch.csv"), "penta_pipe_kernel")
fdm, pm, p8 = lm["fd_kernel"], lm["penta_pipe_kernel"], lm["penta_pipe8_predicted"]
traffic = (tr["penta_pipe_kernel"]["fetch_bytes_per_launch_x2"] + tr["penta_pipe_kernel"]["write_bytes_per_launch_raw"]) / 1e6
gbs = 2259440 / pipe_us / 1e3
cpu = b["cpu_baseline"]
m = lambda pat: re.search(pat, cfgs)
tl = [l.rstrip() for l in open(P("nd_timeline_gn_step.txt")) if re.match(r"(P0|P3|J1|J2|separator  )", l) or "chains saw" in l or "block row 1 part 0" in l]
fi = dict(re.findall(r"(\w+): ([\d.]+) ms/iteration", open(P("full_iteration_times.txt")).read()))
mpc = re.search(r"mini_cheetah: N=20.*?median ([\d.]+) ms, p10 ([\d.]+), p90 ([\d.]+)", open(P("mpc_latency.txt")).read())
numbers = f"""| | |
|---|---|
| Gauss-Newton iterations / s, one problem (`value`) | **{b['value']:.0f}** ({1e3 * b['ms_per_step']:.1f} µs a step; round 5: 12,358, round 4: 11.9k, round 3: 10.5k, round 2: 8.3k, round 1: 6.2k) |
| kernels, rocprofv3 averages (`profiles/{R}_kernel_stats.csv`) | `fd_kernel<3,3>` {fd_us:.2f} µs, `penta_pipe_kernel<19>` {pipe_us:.2f} µs with the assembly inside ({pipe_alone:.2f} alone: `..._assembly_in_its_own_launch.csv`); HIP events {1e3 * b['roofline']['all_kernels_avg_ms']['fd_kernel']:.1f} + {1e3 * b['roofline']['all_kernels_avg_ms']['penta_pipe_kernel']:.1f} |
| one step, synchronised before and after | median {1e3 * b['step_latency_ms']['median']:.1f} µs (p10 {1e3 * b['step_latency_ms']['p10']:.1f}, p90 {1e3 * b['step_latency_ms']['p90']:.1f}) |
| `roofline` (the solver's launch) | {b['roofline']['achieved']:.1f} GB/s algorithmic = {b['roofline']['frac']:.4f} of 8 TB/s; counter traffic {traffic:.2f} MB a launch = {traffic * 1e6 / 2259440:.2f} × the algorithmic bytes |
| CPU port on the box's host cores (`cpu_baseline`, {cpu['iterations_per_repeat']} iterations × {cpu['repeats']}, median) | {", ".join(f"{t} threads: {v:.0f} it/s (spread {100 * (cpu['spread_by_num_threads'].get(t) or 0):.0f} %)" for t, v in cpu['iters_per_s_by_num_threads'].items())}; best leg {cpu['value']:.0f} ⇒ **{b['speedup_vs_cpu_baseline']:.1f} ×** (the ratio moves with the host: the same command gave 209 / 392 / 461 / 486 it/s in `{R}_all_configs.txt`, i.e. {b['value'] / 486:.0f} ×) |
| batch of 16 / 64 problems, one host thread | {b['batch_mode'][0]['value'] / 1e3:.0f}k / {b['batch_mode'][1]['value'] / 1e3:.0f}k it/s aggregate |
| full trust-region iteration (`TrajectoryOptimizer::Solve`, the YAML's settings) | {b['full_iteration']['ms_per_iteration']:.3f} ms (CPU port {b['full_iteration']['cpu_port_ms_per_iteration']:.2f} ms) |
| MPC re-plan (cheetah N = 20, `idto_mpc_update`) | median {b['mpc_replan']['ms_per_replan_median']:.3f} ms |"""
vals = {
    "R6": R, "R6_FD_US": f"{fd_us:.1f}", "R6_PIPE_US": f"{pipe_us:.1f}", "R6_PIPE_ALONE_US": f"{pipe_alone:.1f}",
    "R6_STEP_US": f"{1e3 * b['ms_per_step']:.1f}", "R6_ITS": f"{b['value']:.0f}",
    "R6_FD_VALU": f"{fdm['valu_instructions_per_wavefront']:,}", "R6_FD_WAIT": f"{100 * fdm['wait_frac_of_wave_cycles']:.0f} %",
    "R6_FD_ACTIVE": f"{100 * fdm['valu_active_frac_of_wave_cycles']:.0f} %", "R6_FD_FLOOR": f"{fdm['issue_floor_us']:.1f}",
    "R6_FD_FRAC": f"{fdm['achieved_frac_of_issue_floor']:.2f}",
    "R6_FD44_US": m(r"allegro_hand N=60:.*?fd_kernel ([\d.]+)").group(1),
    "R6_ASM_US": m(r"allegro_hand N=60:.*?assemble_terms_kernel ([\d.]+)").group(1),
    "R6_ND23_US": m(r"allegro_hand N=60:.*?penta_nd_kernel ([\d.]+)").group(1),
    "R6_BAND6_US": "17.5 as a launch of its own",
    "R6_TIMELINE": "\n".join("    " + l for l in tl),
    "R6_PIPE_GBS": f"{gbs:.1f}", "R6_PIPE_FRAC": f"{gbs / 8000:.4f}", "R6_PIPE_TRAFFIC": f"{traffic:.2f}",
    "R6_ROW_MODEL": f"{pm['row_model_us']:.1f}", "R6_PIPE8_FLAT": f"{p8['flat_three_separators_us']:.1f}",
    "R6_PIPE8_NESTED": f"{p8['nested_sub_separators_us']:.1f}", "R6_PIPE8_TODAY": f"{p8['model_of_todays_kernel_us']:.1f}",
    "R6_NUMBERS": numbers, "R6_ALL_CONFIGS": "\n".join("    " + l for l in cfgs.strip().splitlines()),
    "R6_SPEEDUP": f"{b['speedup_vs_cpu_baseline']:.0f} – {b['value'] / 486:.0f}", "R6_SPEEDUP1": f"{b['value'] / 209:.0f}",
    "R6_BATCH64": f"{b['batch_mode'][1]['value'] / 1e3:.0f}k",
    "R6_FULLITER": f"{b['full_iteration']['ms_per_iteration']:.3f}", "R6_FULLITER_CPU": f"{b['full_iteration']['cpu_port_ms_per_iteration']:.1f} ms",
    "R6_MPC": f"{mpc.group(1)} (p10 {mpc.group(2)}, p90 {mpc.group(3)})",
    "R6_TRITER_US": f"{avg_us(P('full_iteration_kernel_stats.csv'), 'tr_iter_kernel'):.1f}",
    "R6_COST_US": f"{avg_us(P('full_iteration_kernel_stats.csv'), 'cost_kernel'):.1f}",
    "R6_SMALL_ITER": small_iter_table(),
    "R6_FULLITER_OTHERS": ", ".join(f"{k} {float(v):.3f}" for k, v in fi.items() if k != "mini_cheetah"),
}
src = os.path.join(ROOT, "tools", "design", "DESIGN.in.md")
text = open(src).read()
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_design.py")])
out = open(os.path.join(ROOT, "DESIGN.md")).read()
for k in sorted(vals, key=len, reverse=True):
    out = out.replace(f"@@{k}@@", vals[k])
left = sorted(set(re.findall(r"@@\w+@@", out)))
if left:
    raise SystemExit("unfilled: " + " ".join(left))
open(os.path.join(ROOT, "DESIGN.md"), "w").write(out)
print("DESIGN.md filled from profiles/%s_*" % R)

"""Timeline of the seven-workgroup solver on the KKT system of a constrained trust-region iteration (option
"solver_debug" 2: the stamps of the KKT context).  python tools/kkt_timeline.py allegro_hand 40"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, SCALING
name, N = sys.argv[1], int(sys.argv[2])
cfg, model = load_config(name), load_model(name)
prob, sp, q_guess = make_problem(cfg, model, num_steps=N)
dev = hip.HipPath(model, prob, sp)
dev.set_unactuated_dofs(model.unactuated_dofs)
for kv in os.environ.get("IDTO_TIMELINE_OPTS", "").split(","):
    if "=" in kv:
        dev.set_option(kv.split("=")[0], int(kv.split("=")[1]))
args = (SCALING[sp.scaling_method] if sp.scaling else -1, sp.scaling, False, sp.Delta0, sp.Delta_max)
dev.set_q(np.asarray(q_guess)); dev.eval_tau()
dev.tr_solve(3, *args, constrained_dofs=model.unactuated_dofs)
for kv in os.environ.get("IDTO_TIMELINE_OPTS_LATE", "").split(","):
    if "=" in kv:
        dev.set_option(kv.split("=")[0], int(kv.split("=")[1]))
dev.set_option("solver_debug", 2)
dev.set_q(np.asarray(q_guess)); dev.eval_tau()
dev.tr_solve(2, *args, constrained_dofs=model.unactuated_dofs)
d = dev.get("debug")[:7 * 64].reshape(7, 64) / 100.0
t0 = d[:4, 0].min()
at = lambda x: f"{x - t0:.2f}" if x > 0 else "-"
print(f"{name} N={N}: KKT solver that ran: {dev.get_option('kkt_last_solver')} (2: seven workgroups)")
names = ["P0 producer", "P3 producer", "J1 joiner", "J2 joiner", "spike J1", "spike J2", "separator"]
for r in range(4):
    x = d[r]
    print(f"{names[r]:12s} start {at(x[0])}  join-wait-begin {at(x[1])}  join-wait-end {at(x[5])}  forward done {at(x[2])}  backward start {at(x[3])}  end {at(x[4])}")
    if r >= 2:
        print("   rows published:", " ".join(at(v) for v in x[24:44] if v > 0))
    if x[22] > 0:
        print(f"   back substitution in recursion form: separator's solution in {at(x[21])}, recursion from {at(x[22])} to {at(x[4])}")
for r in (4, 5):
    x = d[r]
    rows = [(x[8 + 2 * i], x[9 + 2 * i]) for i in range(16) if x[8 + 2 * i] > 0]
    print(f"{names[r]:12s} start {at(x[0])}  last row published {at(x[1])}")
    print("   rows (ready, done):", " ".join(f"({at(a)},{at(b)})" for a, b in rows))
x = d[6]
print("separator: rows of spike workgroup 0 seen at", " ".join(at(v) for v in x[24:48] if v > 0))
print(f"   last row of spike workgroup 0: operands in registers {at(x[50])}, accumulated {at(x[51])}")
print(f"{names[6]:12s} start {at(x[0])}  Q ready {at(x[1])}  W built {at(x[3])}  row s {at(x[4])}  S' {at(x[5])}  row s+1 {at(x[6])}  solved+posted {at(x[2])}")

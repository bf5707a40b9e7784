#!/usr/bin/env python3
"""A/B/C... of environment settings on ONE box, interleaved fresh processes of tools/step_times.py (host-buffer send proofs): per variant the medians over the repetitions of
the step median and of the prover's own device clock (the figure that does not move with the host's mood).
python tools/ab_device.py [--reps 3] [--steps 300] "VAR=a VAR2=b" "VAR=c" ...     ("-" = no variable set)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); args = sys.argv[1:]; reps, steps = 3, 300
while args and args[0].startswith("--"):
    if args[0] == "--reps": reps = int(args[1])
    if args[0] == "--steps": steps = int(args[1])
    args = args[2:]
res = {v: [] for v in args}
for r in range(reps):
    for v in args:
        env = dict(os.environ); env.update(dict(kv.split("=", 1) for kv in v.split() if "=" in kv))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "step_times.py"), str(steps)], env=env, capture_output=True, text=True).stdout
        l = [x for x in out.splitlines() if x.startswith("RESULT")]
        if l: t = l[0].split(); res[v].append((float(t[2]), float(t[6]), float(t[8])))
med = lambda xs: sorted(xs)[len(xs) // 2] if xs else float("nan")
print("%-52s %10s %10s %10s   (medians over %d fresh processes of %d steps; all device clocks: ...)" % ("variant", "step ms", "device ms", "upload ms", reps, steps))
for v in args: print("%-52s %10.4f %10.4f %10.4f   %s" % (v, med([x[0] for x in res[v]]), med([x[1] for x in res[v]]), med([x[2] for x in res[v]]), " ".join("%.3f" % x[1] for x in res[v])))

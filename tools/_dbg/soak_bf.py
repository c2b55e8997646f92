"""All fuzz seeds of one branching factor under BBHIP_SYS=1, `reps` times each; prints failures only."""
import os, sys, subprocess
bf, lo, hi, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
bad = 0; runs = 0
for seed in range(lo, hi):
    if (50 if seed % 3 else 254) != bf: continue
    out = subprocess.run([sys.executable, "/root/repo/tools/_dbg/repeat_seed.py", str(seed), str(reps)], capture_output=True, text=True).stdout
    last = out.strip().splitlines()[-1] if out.strip() else "no output"
    runs += reps
    if " 0 bad of" not in last:
        bad += 1
        print("\n".join(l[:260] for l in out.strip().splitlines()[-6:]), flush=True)
print(f"bf {bf}: seeds {lo}..{hi - 1} x {reps}: {bad} seeds with failures, {runs} runs")

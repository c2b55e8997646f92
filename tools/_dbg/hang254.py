import os, sys
os.environ["BBHIP_SYS"] = "1"; os.environ["BBHIP_SYS_DEBUG"] = "1"
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from bench import WORKLOADS
from bblean_amd import BitBirch
gen, thr, _ = WORKLOADS["zipf"]
fps = gen(1_000_000, 4321, torch.device("cuda"))
for rep in range(int(sys.argv[1])):
    try:
        BitBirch(branching_factor=254, threshold=thr, merge_criterion="diameter").fit(fps)
    except Exception as exc:
        print(f"rep {rep}: {exc!r}"[:400], flush=True)
print("done")

import os, sys
os.environ["BBHIP_SYS"]="1"; os.environ["BBHIP_TINY_POOLS"]="1"; os.environ["BBHIP_LAUNCH_LOG"]="1"
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from bench import WORKLOADS
from bblean_amd import BitBirch
rows = WORKLOADS["hier"][0](30000, 99+254, torch.device("cuda")).cpu().numpy()
t = BitBirch(branching_factor=254, threshold=0.6, merge_criterion="diameter")
t.fit(rows[:10000]); print("stats", t._engine.stats().tolist(), "kc", t._engine.kernel_counts().tolist(), "sys", t._engine.sys_counts().tolist(), "mem", t._engine.memory().tolist())

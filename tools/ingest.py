"""f3: streaming ingest (reference bblean/_memory.py:74-126 memory-maps the .npy and releases pages as it goes).
`BitBirch.fit(path)` from a page-cache-cold .npy against `fit` of the same rows resident in HBM, and the arr-vec
Tanimoto kernel on rows in pinned host memory (the PCIe-inclusive rate - never the bench's `value`).
    python tools/ingest.py [rows]"""
import os, sys, time, tempfile
os.environ.pop("BBHIP_LAUNCH_LOG", None)  # (any value, "0" included, turns the log on)
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import torch
from pathlib import Path
from bench import synth_ecfp
from bblean_amd import BitBirch
from bblean_amd.similarity import _jt_sim_arr_vec_packed

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
dev = torch.device("cuda")
fps = synth_ecfp(n, 77, dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
a = BitBirch(branching_factor=254, threshold=0.3).fit(fps)
torch.cuda.synchronize()
t_hbm = time.perf_counter() - t0
host = fps.cpu().numpy()
with tempfile.TemporaryDirectory(dir="/tmp") as d:
    f = Path(d) / "fps.npy"
    np.save(f, host)
    os.sync()
    cold = False
    try:
        open("/proc/sys/vm/drop_caches", "w").write("3\n")
        cold = True
    except OSError:
        pass
    t0 = time.perf_counter()
    b = BitBirch(branching_factor=254, threshold=0.3).fit(f)
    torch.cuda.synchronize()
    t_file = time.perf_counter() - t0
same = bool((a.get_assignments() == b.get_assignments()).all())
print(f"fit of {n} S-ecfp rows ({n * 256 / 1e9:.2f} GB), bf 254: resident in HBM {t_hbm:.2f} s ({n / t_hbm:.0f} fps/s); "
      f"from a {'page-cache-cold' if cold else 'warm (drop_caches not permitted)'} .npy through HostSlabs {t_file:.2f} s "
      f"({n / t_file:.0f} fps/s, {n * 256 / t_file / 1e6:.0f} MB/s from the file); same clusters: {same}")
pinned = torch.from_numpy(host).pin_memory()
vec = fps[0].clone()
for arr, tag in ((pinned.numpy(), "pinned host rows"), (host, "pageable host rows")):
    _jt_sim_arr_vec_packed(arr[:100000], vec)
    t0 = time.perf_counter()
    _jt_sim_arr_vec_packed(arr, vec)
    dt = time.perf_counter() - t0
    print(f"K1 arr-vec Tanimoto on {tag}: {dt * 1e3:.1f} ms = {n * 256 / dt / 1e9:.1f} GB/s PCIe-inclusive ({n / dt / 1e6:.1f} M rows/s)")

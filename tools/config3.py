"""BASELINE.json configs[2]: S-ecfp(N) (sparse ECFP4-like rows around N/50 planted prototypes,
SURVEY.md section 8d), threshold 0.3, `bb run --refine-num 1` sequence (cli.py:1067-1092) with the CLI's default
branching factor 254 (or the one given):
fit -> set_merge(tolerance-diameter, tol 0.05) -> refine_inplace(n_largest=1).

    python tools/config3.py N [check_n] [branching_factor]

Times the HIP engine on N rows; when check_n > 0 the first check_n rows are also run through
the CPU oracle with the same host logic and the cluster ids compared."""
import sys
import time

sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import torch

BF = 254  # the CLI's default; main() takes another one from the command line


def synth_ecfp(n: int, seed: int, device, n_features: int = 2048):
    g = torch.Generator(device=device).manual_seed(seed)
    k = max(n // 50, 1)
    pops = torch.clamp(torch.round(torch.randn(k, device=device, generator=g) * 12.0 + 48.0), 8, 160).to(torch.int64)
    weights = torch.tensor([128, 64, 32, 16, 8, 4, 2, 1], dtype=torch.int32, device=device)
    protos = torch.empty((k, n_features // 8), dtype=torch.uint8, device=device)
    chunk = 50_000
    for lo in range(0, k, chunk):
        m = min(chunk, k - lo)
        ranks = torch.rand((m, n_features), device=device, generator=g).argsort(dim=1).argsort(dim=1)
        bits = (ranks < pops[lo:lo + m, None]).to(torch.int32)
        protos[lo:lo + m] = (bits.view(m, -1, 8) * weights).sum(dim=2).to(torch.uint8)
    out = torch.empty((n, n_features // 8), dtype=torch.uint8, device=device)
    shifts = torch.arange(7, -1, -1, device=device, dtype=torch.uint8)
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        which = torch.randint(0, k, (m,), device=device, generator=g)
        pb = ((protos[which][:, :, None] >> shifts) & 1).bool().view(m, n_features)
        keep = torch.rand((m, n_features), device=device, generator=g) > 0.15
        add = torch.rand((m, n_features), device=device, generator=g) < (0.15 * pops[which].double() / n_features)[:, None]
        bits = ((pb & keep) | add).to(torch.int32)
        out[lo:lo + m] = (bits.view(m, -1, 8) * weights).sum(dim=2).to(torch.uint8)
    return out


def prof(lib, tag):
    import ctypes as C
    out = []
    for name in (b"tree_insert", b"gather_leaves"):
        l, ms = C.c_int64(0), C.c_double(0.0)
        lib.bbh_profile_get(name, C.byref(l), C.byref(ms))
        out.append(f"{name.decode()} {ms.value / 1e3:.2f}s/{l.value}")
    lib.bbh_profile_reset()
    return f"[{tag}: " + ", ".join(out) + "]"


def run(fps, host, engine_factory=None):
    from bblean_amd import BitBirch, _lib
    lib = _lib.load()
    lib.bbh_profile_enable(1)
    lib.bbh_profile_reset()
    kw = {} if engine_factory is None else {"_engine_factory": engine_factory}
    t0 = time.perf_counter()
    tree = BitBirch(branching_factor=BF, threshold=0.3, merge_criterion="diameter", **kw)
    tree.fit(fps)
    t1 = time.perf_counter()
    k_fit = len(tree._leaves()["ids"])
    if engine_factory is None:
        print(prof(lib, "fit"), tree._engine.stats(), flush=True)
    tree.set_merge("tolerance-diameter", tolerance=0.05, threshold=0.3)
    tree.refine_inplace(host, n_largest=1)
    t2 = time.perf_counter()
    if engine_factory is None:
        print(prof(lib, "refine"), tree._engine.stats(), flush=True)
    ids = tree.get_assignments()
    t3 = time.perf_counter()
    return ids, dict(fit=t1 - t0, refine=t2 - t1, assign=t3 - t2, clusters_after_fit=k_fit, clusters=int(ids.max()))


def main():
    global BF
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    check_n = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    BF = int(sys.argv[3]) if len(sys.argv) > 3 else BF
    dev = torch.device("cuda")
    fps = synth_ecfp(n, 7, dev)
    torch.cuda.synchronize()
    host = fps.cpu().numpy()
    ids, t = run(fps, host)
    print(f"HIP  N={n}: fit {t['fit']:.2f}s ({n / t['fit']:.0f} fps/s)  refine {t['refine']:.2f}s  "
          f"clusters {t['clusters_after_fit']} -> {t['clusters']}  end-to-end {n / (t['fit'] + t['refine']):.0f} fps/s", flush=True)
    if check_n:
        from oracle_engine import OracleEngine
        hs = host[:check_n]
        ids_h, th = run(fps[:check_n], hs)
        ids_o, to = run(hs, hs, engine_factory=OracleEngine)
        same = bool(np.array_equal(ids_h, ids_o))
        print(f"check N={check_n}: HIP fit {th['fit']:.2f}s refine {th['refine']:.2f}s | oracle(1 core) fit {to['fit']:.2f}s "
              f"refine {to['refine']:.2f}s | clusters {th['clusters']} vs {to['clusters']} | identical ids: {same}", flush=True)
        if not same:
            sys.exit(1)


if __name__ == "__main__":
    main()

r"""Compile-time guards on the steady-state insertion kernel (no GPU needed: hipcc cross-compiles gfx950).

Two regressions of this kind cost 5-10 % each before they were found in the ISA: per-level
state that the optimiser left in private memory (a scratch load + `s_waitcnt vmcnt(0)` in every level of every
descent), and a call inside the insertion loop.  The test compiles the device code of bb_tree.hip to assembly and
checks every `k_tree_fast` instance: no scratch instruction in its body, and a private segment no larger than what
the out-of-line cold functions it calls need."""
from __future__ import annotations

import re
import shutil
import subprocess
from pathlib import Path

import pytest

CSRC = Path(__file__).resolve().parents[1] / "bblean_amd" / "csrc"
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="hipcc not available")
def test_fast_kernels_use_no_scratch_memory(tmp_path):
    out = tmp_path / "bb_tree.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-ffp-contract=off", "-Wno-unused-function",
           "--cuda-device-only", "-S", str(CSRC / "bb_tree.hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    text = out.read_text()
    kernels = re.findall(r"^(_ZN\S*k_tree_fast\S*):", text, re.M)
    assert len(kernels) >= 10, kernels  # bf 50 / 254 x packed / buffers x criteria (+ the phase-timer builds)
    for name in kernels:
        if "Lb1EEE" in name:  # PROF = true: the phase-timer builds (BBHIP_PHASES) keep their timer array in memory
            continue
        start = text.index(name + ":")
        body = text[start:text.index(".Lfunc_end", start)]
        scratch = [ln.strip() for ln in body.splitlines() if ln.strip().startswith(("scratch_", "buffer_load", "buffer_store"))]
        assert not scratch, (name, scratch[:5])
        # private segment: "<own bytes> + max(<callees>)" - the kernel's own part must be zero
        m = re.search(r"\.set " + re.escape(name) + r"\.private_seg_size, (\d+)\+max\(", text)
        assert m is not None and int(m.group(1)) == 0, (name, m.group(0) if m else None)
        # and no call inside the steady-state loop: the only calls are the two cold functions of tree_fast_body
        calls = len(re.findall(r"s_swappc_b64", body))
        assert calls <= 2, (name, calls)
    # the pipelined kernel (bb_tree_pipe.inc): four specialised waves whose loop-carried state lives in registers and LDS -
    # nothing of it may end up in private memory either, and its only calls are the two cold functions between runs
    pipes = re.findall(r"^(_ZN\S*k_tree_pipe\S*):", text, re.M)
    assert len(pipes) >= 4, pipes  # bf 50 / 254 x diameter / tolerance-diameter (+ the phase-timer builds)
    for name in pipes:
        if "Lb1EEE" in name:  # PROF = true (BBHIP_PIPE_PHASES): per-kind timers indexed at run time live in memory
            continue
        start = text.index(name + ":")
        body = text[start:text.index(".Lfunc_end", start)]
        scratch = [ln.strip() for ln in body.splitlines() if ln.strip().startswith(("scratch_", "buffer_load", "buffer_store"))]
        assert not scratch, (name, scratch[:5])
        m = re.search(r"\.set " + re.escape(name) + r"\.private_seg_size, (\d+)\+max\(", text)
        assert m is not None and int(m.group(1)) == 0, (name, m.group(0) if m else None)
        assert len(re.findall(r"s_swappc_b64", body)) <= 2, name

"""Shader clock and socket power while one GEMM variant runs back to back (rocm-smi sampled from a side thread)."""
import ctypes as C
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ramp_amd import _lib

lib = _lib.load_tools()
samples = []
stop = False


def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "-c", "-P"], capture_output=True, text=True, timeout=20).stdout
        except Exception as e:  # noqa: BLE001
            o = str(e)
        sc = re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", o)
        pw = re.findall(r"Power \(W\): ([\d.]+)", o)
        samples.append((time.time(), sc[:1], pw[:1]))


def run(label, M, N, K, mode, flags, secs=4.0, flops=None):
    us = C.c_float()
    t0 = time.time(); n0 = len(samples); last = 0.0
    while time.time() - t0 < secs:
        _lib.check_tools(lib.ramp_bench_gemm(M, N, K, 1, 1, mode, flags, 1, 200, C.byref(us), None))
        last = us.value
    got = samples[n0:]
    sc = [int(s[1][0]) for s in got[len(got) // 2:] if s[1]]; pw = [float(s[2][0]) for s in got[len(got) // 2:] if s[2]]
    fl = flops if flops is not None else 2.0 * M * N * K
    print(f"{label}: {last:.1f} us ({fl / last / 1e6:.0f} TF)  sclk {sum(sc) / max(1, len(sc)):.0f} MHz  "
          f"power {sum(pw) / max(1, len(pw)):.0f} W  ({len(sc)} samples)", flush=True)


th = threading.Thread(target=sampler, daemon=True); th.start()
time.sleep(2.0)
print("idle", samples[-2:], flush=True)
M = 393216
run("full 256x256", M, 256, 256, 3, 3)
run("mfma+lds-reads only 256x256", M, 256, 256, 3, 3 | (15 << 8))
run("no-epilogue 256x256", M, 256, 256, 3, 3 | (8 << 8))
run("no A staging 256x256", M, 256, 256, 3, 3 | (1 << 8))
run("no global stores 256x256", M, 256, 256, 3, 3 | (16 << 8))
run("full 768x256", M, 768, 256, 3, 1)
run("full 256x768", M, 256, 768, 3, 3)
run("full GEGLU 2048x256", M, 2048, 256, 3, 1 | 4)
run("fused FF", M, 2048, 256, 5, 0)
run("full 2048x2048", 65536, 2048, 2048, 3, 1)
run("mfma+lds-reads only 2048x2048", 65536, 2048, 2048, 3, 1 | (15 << 8))
run("bf16x6 2048x2048", 65536, 2048, 2048, 1, 1)
run("fp32 2048x2048", 65536, 2048, 2048, 0, 1)
ff = 2.0 * M * (2048 * 256 + 256 * 1024)
run("token-owning fused feed-forward, forward (ffx)", M, 2048, 256, 6, 0, flops=ff)
run("token-owning fused feed-forward, backward (ffx)", M, 2048, 256, 7, 0, flops=ff)
run("ffx forward, no stash traffic", M, 2048, 256, 6, 2 << 8, flops=ff)
run("ffx forward, no weight DMA", M, 2048, 256, 6, 1 << 8, flops=ff)
run("token-owning LN -> QKV (tkl)", M, 768, 256, 8, 1)
run("tkl LN -> QKV, no DMA, no stores", M, 768, 256, 8, 1 | (3 << 8))
stop = True

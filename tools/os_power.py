"""Core clock and package power while ONE GEMM geometry runs in a tight HIP-graph loop (is the kernel power / clock limited?):
python tools/os_power.py <mode> <which: auto|os2|os3|64x128|256x128> [seconds]; the library is $ALDM_LIB_PATH or the shipped one."""
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm2_amd import ops  # noqa: E402

ops.set_mma(sys.argv[1])
from tools.os_probe import Case  # noqa: E402

which = sys.argv[2]
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
force = {"auto": None, "os2": (32, 128, 302), "os3": (32, 128, 303), "os4": (32, 128, 304), "64x128": (64, 128, 2), "128x128": (128, 128, 2),
         "256x128": (256, 128, 2)}[which]
c = Case("geglu 16384x256->2x1024", "geglu", 16 * 1024, 256, 2048, split_out="only")
c.run(force)
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for _ in range(50):
        c.run(force)
samples = []
stop = False


def sampler():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
        m = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
        w = re.search(r"Power \(W\): ([0-9.]+)", out)
        if m and w:
            samples.append((int(m.group(1)), float(w.group(1))))
        time.sleep(0.2)


th = threading.Thread(target=sampler)
th.start()
t0 = time.time()
n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < secs:
    for _ in range(20):
        gr.replay()
    n += 20 * 50
    torch.cuda.synchronize()
e1.record()
torch.cuda.synchronize()
stop = True
th.join()
us = e0.elapsed_time(e1) * 1e3 / n
s = sorted(samples[2:] or samples)
tag = os.path.basename(os.environ.get("ALDM_LIB_PATH", "shipped"))
if s:
    mid = s[len(s) // 2]
    print(f"{sys.argv[1]} {which:8s} {tag:28s} {us:7.1f} us/launch  sclk median {mid[0]} MHz (min {s[0][0]} max {s[-1][0]})  "
          f"power median {sorted(p for _, p in s)[len(s) // 2]:.0f} W  ({len(s)} samples)", flush=True)
else:
    print(f"{sys.argv[1]} {which} {tag} {us:.1f} us/launch (no smi samples)", flush=True)

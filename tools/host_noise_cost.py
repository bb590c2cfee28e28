"""Host cost of RNG contract R under prompt sharding: every rank draws the GLOBAL batch's noise from the host generator and
keeps its rows (audioldm2_amd/ddim.py host_drawer).  Prints ms per DDIM step for one rank at global batch gB (8 prompts per
rank) — to be compared with the GPU's UNet step (18 ms at 8 prompts/GPU): the draw runs on the launching thread in 8-step
chunks between graph replays, so it must stay well below the step time.  CPU only."""
import sys
import time

import torch

sys.path.insert(0, ".")
from audioldm2_amd.ddim import host_drawer  # noqa: E402

torch.manual_seed(0)
for gB in (8, 16, 32, 64):
    shape = (8, 8, 256, 16)
    d = host_drawer(shape, None if gB == 8 else (gB, 8))
    dst = torch.empty(shape)
    for _ in range(3):
        d.into(dst)
    t0 = time.time()
    n = 40
    for _ in range(n):
        d.into(dst)
    ms = (time.time() - t0) / n * 1e3
    print(f"global batch {gB:3d} ({gB // 8} ranks x 8 prompts): {ms:6.2f} ms per step per rank "
          f"({gB * 8 * 256 * 16 / 1e6:.2f} M normals), {ms * 200 / 1e3:.2f} s per 200-step job", flush=True)

# ---- round 3: the draws run on a DRAWER THREAD (ddim._NoiseFeed); what is left on the launching thread per step is a
# condition-variable check + (once per 8 steps) a stream wait.  Emulated here with the real feed on the GPU: the launching
# thread "replays" a 16 ms step (sleep) 200 times and we time what it spends inside feed.wait().  GPU box only.
if torch.cuda.is_available():
    from audioldm2_amd.ddim import _NoiseFeed
    dev = torch.device("cuda")
    for gB in (8, 64):
        shape = (8, 8, 256, 16)
        for threaded in (False, True):
            torch.manual_seed(0)
            d = host_drawer(shape, None if gB == 8 else (gB, 8))
            feed = _NoiseFeed(d, 200, shape, False, 1.0, dev, threaded=threaded)
            feed.produce_next()
            spent = 0.0
            t_all = time.time()
            for i in range(200):
                t0 = time.time()
                opens = feed.wait(i)
                if opens:
                    feed.produce_next()
                spent += time.time() - t0
                time.sleep(0.016)
            feed.close()
            torch.cuda.synchronize()
            print(f"global batch {gB:3d}, drawer thread {'on ' if threaded else 'off'}: launching thread spends "
                  f"{spent / 200 * 1e3:6.3f} ms per step in the noise feed (200 steps, 16 ms emulated GPU step; wall "
                  f"{time.time() - t_all:.2f} s)", flush=True)

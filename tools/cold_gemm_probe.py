"""How much of a short GEMM's in-step duration is cold caches?  In the replayed UNet step every GEMM reads weights that were last
touched one step (1.4 GB of other weights) ago and activations another kernel just wrote; the tuner's graph-timed loops re-run one
launch on L2-resident operands.  Here each launch of the loop is preceded by a kernel that overwrites a scratch buffer (64 MB: more
than the 8 x 4 MB L2s; 640 MB: more than L2 + the 256 MB Infinity Cache), and the scratch kernel alone is timed and subtracted.
Usage (GPU box): python tools/cold_gemm_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = sys.argv[:1]
from tools.ws_probe import Case  # noqa: E402


def graph_time(fn, reps=20, replays=4):
    fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            fn()
    gr.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(replays):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


def main():
    R = 16
    cases = [
        Case("L3 proj 1024x640->640 +res", "linear", R * 64, 640, 640, res=True),
        Case("L3 qkv 1024x640->1920", "linear", R * 64, 640, 1920),
        Case("L2 proj 4096x384->384 +res", "linear", R * 256, 384, 384, res=True),
        Case("L2 ffout 4096x1536->384 +res", "linear", R * 256, 1536, 384, res=True),
        Case("L1 proj 16384x256->256 +res", "linear", R * 1024, 256, 256, res=True),
        Case("L1 qkv 16384x256->768", "linear", R * 1024, 256, 768),
        Case("L1 geglu 16384x256->2x1024 (split out)", "geglu", R * 1024, 256, 2048, split_out="only"),
        Case("L1 conv3x3 256->256 @128x8 +res", "conv3", R * 1024, 2304, 256, res=True, hw=(R, 128, 8)),
    ]
    scratch = {mb: torch.empty(mb * 1024 * 1024 // 4, device="cuda") for mb in (64, 640)}
    t_flush = {mb: graph_time(lambda: scratch[mb].fill_(1.0), reps=10) for mb in scratch}
    print("# us per launch, HIP-graph timed; 'after 64 MB' / 'after 640 MB': each launch preceded by a kernel overwriting that much "
          f"scratch (its own time, {t_flush[64]:.1f} / {t_flush[640]:.1f} us, subtracted)", flush=True)
    for c in cases:
        hot = graph_time(lambda: c.run())
        line = f"{c.name:44s} L2-hot {hot:6.1f}"
        for mb in (64, 640):
            def both():
                scratch[mb].fill_(1.0)
                c.run()
            t = graph_time(both, reps=10) - t_flush[mb]
            line += f" | after {mb} MB {t:6.1f} (+{t - hot:4.1f})"
        print(line, flush=True)


if __name__ == "__main__":
    main()

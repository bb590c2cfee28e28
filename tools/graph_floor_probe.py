"""What does a dependent chain of N launches cost in a HIP-graph replay when the kernels do (almost) nothing?  The UNet step is 856
dependent launches; this is the floor under it.  Chains of (a) a 1-block kernel, (b) a 256-block x 256-thread elementwise kernel on
64 K floats, (c) a 1024-block one on 1 M floats (4 MB in, 4 MB out) — time per node.  Usage (GPU box): python tools/graph_floor_probe.py"""
import torch

N = 856


def chain_time(fn, n=N, replays=5):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(replays):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def main():
    for name, numel in (("1 block (64 floats)", 64), ("256 blocks (64 K floats)", 65536), ("4096 blocks (1 M floats)", 1 << 20),
                        ("16 M floats (64 MB in, 64 MB out)", 1 << 24)):
        x = torch.zeros(numel, device="cuda")
        y = torch.empty_like(x)

        def fn():
            torch.add(x, 1.0, out=y)
        ms = chain_time(fn)
        print(f"chain of {N} dependent launches, {name:36s}: {ms:7.3f} ms per replay = {ms / N * 1e3:6.2f} us per node", flush=True)


if __name__ == "__main__":
    main()

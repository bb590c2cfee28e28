"""DESIGN §9.0 / VERDICT r2 next #7: the six batch-16 attention parity cases of tests/test_parity_configs_gpu.py, N times in
ONE process — once plainly, once with a cyclic garbage collection forced between launches and HIP-graph captures interleaved
(the round-2 abort was traced to a collection running inside a graph capture; this is the stress form of that) — counting
wrong results.  A process abort ends the run: the last line printed names the iteration.  Usage: python tools/attn_repeat.py [N]"""
import gc
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_parity_configs_gpu as T  # noqa: E402
from audioldm2_amd import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
CASES = [(1024, False, 1), (512, True, 1), (1024, False, 2), (512, True, 2), (1024, False, 3), (512, True, 3)]
# references once (CPU einsum is the slow part); then only the GPU side repeats
refs = {}
for Lk, masked, mode in CASES:
    refs[(Lk, masked)] = None
B, heads, Lq = 16, 8, 1024
Cc = heads * 32
inputs = {}
for Lk, masked in refs:
    gen = torch.Generator().manual_seed(7)
    q = torch.randn(B, Lq, Cc, generator=gen)
    kv = torch.randn(B, Lk, 2 * Cc, generator=gen)
    mask = None
    if masked:
        mask = (torch.rand(B, Lk, generator=gen) > 0.25).float()
        mask[:, 0] = 1
    ref = T._ref_attention(q, kv[:, :, :Cc].contiguous(), kv[:, :, Cc:].contiguous(), heads, mask)
    inputs[(Lk, masked)] = (q.cuda(), kv.cuda(), None if mask is None else mask.cuda(), ref.double())
bad = 0
worst = 0.0
for phase in ("plain", "gc + graph captures"):
    for it in range(N):
        for Lk, masked, mode in CASES:
            q, kv, mask, ref = inputs[(Lk, masked)]
            prev = ops.attention_mma(mode)
            try:
                if phase != "plain" and it % 10 == 0:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        y = ops.attention(q, kv[:, :, :Cc], kv[:, :, Cc:], heads, mask=mask)
                    g.replay()
                    del g
                    gc.collect()
                else:
                    y = ops.attention(q, kv[:, :, :Cc], kv[:, :, Cc:], heads, mask=mask)
            finally:
                ops.attention_mma(prev)
            if it % 20 == 0 or it == N - 1:   # check a sample of the results (the D2H + fp64 compare dominates otherwise)
                e = float((y.double().cpu() - ref).abs().max() / ref.abs().max())
                worst = max(worst, e)
                if not e < 5e-5:
                    bad += 1
                    print(f"WRONG RESULT phase {phase} iteration {it} case {(Lk, masked, mode)}: rel err {e:.3e}", flush=True)
        if it % 50 == 0:
            print(f"{phase}: iteration {it} ok", flush=True)
    torch.cuda.synchronize()
print(f"attention parity cases: {2 * N} x 6 launches in one process, {bad} wrong results, worst sampled rel err {worst:.2e}, no abort")

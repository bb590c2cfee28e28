"""Per-launch time of the decode step's kernels at GPT-2 base geometry, batch 8, HIP-graph timed, with a different weight
tensor per launch (12 "layers": the weights arrive from HBM, as in the real decode step).  ALDM_LIB_PATH selects a variant."""
import sys
import time

import torch

sys.path.insert(0, ".")
from audioldm2_amd import ops  # noqa: E402

B, E, L = 8, 768, 12
dev = "cuda"
g = torch.Generator().manual_seed(0)


def graph_time(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn()
    gr.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        gr.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


flush = torch.empty(96 << 20, device=dev)   # 384 MB written between shapes: nothing of the last shape stays cached
for name, K, N, ln, act, res in (("c_attn  768->2304 +LN", 768, 2304, True, ops.ACT_NONE, False),
                                 ("c_proj  768->768 +res", 768, 768, False, ops.ACT_NONE, True),
                                 ("c_fc    768->3072 +LN +gelu_new", 768, 3072, True, ops.ACT_GELU_TANH, False),
                                 ("m_proj 3072->768 +res", 3072, 768, False, ops.ACT_NONE, True)):
    ws = [(torch.randn(K, N, generator=g) / K ** 0.5).to(dev) for _ in range(L)]
    bs = [torch.randn(N, generator=g).to(dev) for _ in range(L)]
    x = torch.randn(B, K, generator=g).to(dev)
    r = torch.randn(B, N, generator=g).to(dev) if res else None
    lnp = (torch.ones(K, device=dev), torch.zeros(K, device=dev), 1e-5) if ln else None

    def run():
        for w, b in zip(ws, bs):
            ops.decode_linear(x, w, b, ln=lnp, act=act, res=r)
    flush.fill_(1.0)
    t = graph_time(run) / L * 1e6
    mb = K * N * 4 / 1e6
    print(f"decode_linear {name:34s}: {t:6.1f} us per launch  ({mb:5.1f} MB of weights: {mb / t:5.2f} TB/s)", flush=True)
    del ws, bs
for n_tot, pos in ((828, 400), (828, 826)):
    kc = [torch.randn(B * 12, n_tot, 64, generator=g).to(dev) for _ in range(L)]
    vc = [torch.randn(B * 12, n_tot, 64, generator=g).to(dev) for _ in range(L)]
    qkv = torch.randn(B, 3 * E, generator=g).to(dev)
    km = torch.ones(B, n_tot, device=dev)
    p = torch.tensor([pos], device=dev)

    def run():
        for k, v in zip(kc, vc):
            ops.decode_attention(qkv, p, k, v, km, 12)
    flush.fill_(1.0)
    t = graph_time(run) / L * 1e6
    print(f"decode_attention B=8 x 12 heads, {pos + 1} of {n_tot} cache slots live: {t:6.1f} us per launch", flush=True)

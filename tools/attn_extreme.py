"""Attention under extreme logits: a key planted in a LATER tile whose score lies `jump` log2 units (per unit of q.u) above / below
everything the first tile held.  For the pre-split kernel selected by $ALDM_ATTN_SCHED and for the fp32-K/V path: non-finite outputs,
error vs the fp64 softmax of the same fp32 q / k / v, and the difference between the two.
A second section plants ONE key per chosen tile maximum in a single query row; `map` as an argument adds the position map (one key
200 log2 units up at every (tile, in-tile index): 256 small launches, minutes) that located the half-row maximum of round 5.
Usage: [ALDM_ATTN_SCHED=0|2] [ALDM_MMA=bf16x3] python tools/attn_extreme.py [map]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm2_amd import ops  # noqa: E402

tag = f"sched={os.environ.get('ALDM_ATTN_SCHED', '1')} {ops.MMA_MODE}"
for B, L, heads in [(16, 128, 8), (2, 128, 2)]:
    for jump in (45.0, 70.0, 120.0, 200.0):
        C = heads * 32
        g = lambda i: torch.Generator().manual_seed(i)
        q, k, v = (torch.randn(B, L, C, generator=g(i)) for i in (1, 2, 3))
        u = torch.randn(heads, 32, generator=g(4))
        u = u / u.norm(dim=-1, keepdim=True)
        alpha = jump * math.sqrt(32.0) / 1.4426950408889634
        for pos, f in [(40, 1.0), (100, 1.5), (127, 1.5)]:
            k[:, pos, :] = (alpha * f * u).reshape(1, C)
        xs = ops.split_rows(torch.cat([q, k, v], -1).cuda())
        pw = ops.pack_conv(torch.eye(3 * C))
        qi, kimg, vtimg = ops.linear_qkv(xs, pw, heads, L)
        qkv = ops.linear(xs, pw)
        a_new = ops.attention_presplit(qi, kimg, vtimg, heads)
        a_old = ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads)
        a_oldc = ops.attention(qkv[..., :C].contiguous(), qkv[..., C:2 * C].contiguous(), qkv[..., 2 * C:].contiguous(), heads)
        sh = lambda t: t.double().cpu().view(B, L, heads, 32).transpose(1, 2)
        qd, kd, vd = (sh(qkv[..., i * C:(i + 1) * C].contiguous()) for i in range(3))
        sc = qd @ kd.transpose(-1, -2) / math.sqrt(32.0)
        ref = (torch.softmax(sc, -1) @ vd).transpose(1, 2).reshape(B, L, C)
        err = lambda a: float((a.double().cpu() - ref).abs().max() / ref.abs().max())
        nf = lambda a: int((~torch.isfinite(a)).sum())
        bad = (~torch.isfinite(a_old)).any(-1).nonzero()
        where = ""
        if len(bad):
            b0, r0 = (int(x) for x in bad[0])
            h0 = int((~torch.isfinite(a_old[b0, r0])).nonzero()[0]) // 32
            row = sc[b0, h0, r0] * 1.4426950408889634
            where = (f"  first bad fp32-K/V row (b {b0}, q {r0}, head {h0}): scores log2 max {float(row.max()):.1f} min {float(row.min()):.1f}"
                     f" tile maxima {[round(float(row[32 * i:32 * i + 32].max()), 1) for i in range(L // 32)]}")
        print(f"{tag}: {B} x {heads} x {L}, jump {jump:5.1f}, max |score| {float(sc.abs().max()) * 1.4427:7.1f} log2: non-finite pre-split {nf(a_new)}"
              f" fp32-K/V {nf(a_old)} (contiguous {nf(a_oldc)});  err vs fp64 pre-split {err(a_new):.2e} fp32-K/V {err(a_old):.2e}{where}", flush=True)


# ---- one query row with chosen scores: which step of the online softmax breaks? ------------------------------------------------
# B = 1, two heads (the QKV epilogue's column tiles need C >= 64; head 1 is all zeros), 128 keys: every query of head 0 = e0, key j =
# s_j * e0 (scores s_j in log2 units, exact), all keys 0 except the planted ones
print("single-row cases: tile maxima (log2 units) -> non-finite outputs of the fp32-K/V path / the pre-split kernel, error vs fp64", flush=True)
CASES = [(254.7, 382.1), (133.0, 199.5), (254.7, 300.0), (254.7, 380.0), (254.7, 381.5), (254.7, 383.0), (300.0, 300.0),
         (0.0, 382.0), (0.0, 130.0), (129.0, 258.0), (100.0, 227.4), (200.0, 327.4), (50.0, 177.4), (50.0, 170.0), (382.0, 382.0),
         (126.0, 126.0), (128.0, 128.0), (100.0, 225.0), (100.0, 226.5), (100.0, 228.0), (100.0, 249.0), (100.0, 251.0)]
for m1, m3 in CASES:
    L, C, H = 128, 64, 2
    sl = torch.zeros(L)
    sl[40], sl[100] = m1, m3
    q = torch.zeros(1, L, C)
    q[..., 0] = 1.0
    k = torch.zeros(1, L, C)
    k[0, :, 0] = sl * math.sqrt(32.0) / 1.4426950408889634
    v = torch.randn(1, L, C, generator=torch.Generator().manual_seed(5))
    xs = ops.split_rows(torch.cat([q, k, v], -1).cuda())
    pw = ops.pack_conv(torch.eye(3 * C))
    qi, kimg, vtimg = ops.linear_qkv(xs, pw, H, L)
    qkv = ops.linear(xs, pw)
    a_new = ops.attention_presplit(qi, kimg, vtimg, H)
    a_old = ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], H)
    sh = lambda t: t.double().cpu().view(1, L, H, 32).transpose(1, 2)
    qd, kd, vd = (sh(qkv[..., i * C:(i + 1) * C].contiguous()) for i in range(3))
    ref = (torch.softmax(qd @ kd.transpose(-1, -2) / math.sqrt(32.0), -1) @ vd).transpose(1, 2).reshape(1, L, C)
    err = lambda a: float((a.double().cpu() - ref).abs().max() / ref.abs().max())
    nf = lambda a: int((~torch.isfinite(a)).sum())
    print(f"{tag}: maxima [0, {m1}, 0, {m3}]: fp32-K/V non-finite {nf(a_old)} err {err(a_old):.2e} first row {a_old[0, 0, :3].tolist()};  pre-split non-finite {nf(a_new)} err {err(a_new):.2e}",
          flush=True)


# ---- where must the large key sit for the running maximum to miss it?  one key at +200 log2 units, every (tile, index in tile) ------
if "map" not in sys.argv[1:]:
    sys.exit(0)
print("position map: one key 200 log2 units above the rest; per key tile the in-tile indices (0..31) whose output is non-finite", flush=True)
L, C, H = 128, 64, 2
pw = ops.pack_conv(torch.eye(3 * C))
for path in ("fp32-K/V", "pre-split"):
    for tile in range(L // 32):
        bad = []
        for idx in range(32):
            q = torch.zeros(1, L, C)
            q[..., 0] = 1.0
            k = torch.zeros(1, L, C)
            k[0, 32 * tile + idx, 0] = 200.0 * math.sqrt(32.0) / 1.4426950408889634
            v = torch.randn(1, L, C, generator=torch.Generator().manual_seed(5))
            xs = ops.split_rows(torch.cat([q, k, v], -1).cuda())
            if path == "pre-split":
                qi, kimg, vtimg = ops.linear_qkv(xs, pw, H, L)
                a = ops.attention_presplit(qi, kimg, vtimg, H)
            else:
                qkv = ops.linear(xs, pw)
                a = ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], H)
            if not bool(torch.isfinite(a).all()):
                bad.append(idx)
        print(f"{tag}: {path:9s} key tile {tile}: non-finite for in-tile indices {bad}", flush=True)

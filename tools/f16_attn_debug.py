"""Debug probe: fp16 K / V^T images of the QKV epilogue, operand-stationary vs classic kernel, against fp64."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm2_amd import ops
ops.set_mma("f16x3")
g = lambda s: torch.Generator().manual_seed(s)
B, L, heads, gain = 2, 256, 8, 3.0
C = heads * 32
x = torch.randn(B, L, C, generator=g(1))
ga, be = torch.ones(C), torch.zeros(C)
wq, wk, wv = (gain * torch.randn(C, C, generator=g(4 + i)) / math.sqrt(C) for i in range(3))
pw = ops.pack_conv(torch.cat([wq, wk, wv], 0))
n = ops.layernorm(x.cuda(), ga.cuda(), be.cuda(), 1e-5, split_out="only")
xn = torch.nn.functional.layer_norm(x.double(), (C,), ga.double(), be.double(), 1e-5)
sh = lambda t: t.view(B, L, heads, 32).transpose(1, 2)
qd, kd, vd = xn @ wq.double().t(), xn @ wk.double().t(), xn @ wv.double().t()
ref = (torch.softmax(sh(qd) @ sh(kd).transpose(-1, -2) / math.sqrt(32), -1) @ sh(vd)).transpose(1, 2).reshape(B, L, C)
for form in [None, (32, 128, 303), (64, 128, 4), (128, 128, 2)]:
    for f16 in (True, False):
        ops.F16_ATTN = f16
        if form:
            ops.igemm_force(form[0], form[1], 1, 0, form[2])
        try:
            q, kimg, vtimg = ops.linear_qkv(n, pw, heads, L)
        finally:
            ops.igemm_force(0, 0, 0)
        a = ops.attention_presplit(q, kimg, vtimg, heads)
        if f16:
            qs, ks, vs = kimg._aldm_f16
            kf = kimg.view(torch.float16).double()
            kf = ((kf[:, :, 0] + kf[:, :, 1]) / ks).reshape(B, L, C).cpu()
        else:
            kp = (kimg.to(torch.int32) << 16).view(torch.float32).double()
            kf = (kp[:, :, 0] + kp[:, :, 1] + kp[:, :, 2]).reshape(B, L, C).cpu()
        ek = (kf - kd).abs()
        print(f"form {form} f16_attn {f16}: attn err {float((a.double().cpu() - ref).abs().max() / ref.abs().max()):.2e}  "
              f"q err {float((q.double().cpu() - qd).abs().max() / qd.abs().max()):.2e}  K image err max {float(ek.max() / kd.abs().max()):.2e} "
              f"rms {float(ek.pow(2).mean().sqrt() / kd.abs().max()):.2e}", flush=True)
ops.F16_ATTN = True
# ---- V^T image decode: [B, heads, tiles, 2, 32 d, 32 slots]; slot c -> s = c >> 4, lh = (c >> 3) & 1, e = 8 s + (c & 7), key = (e & 3) + 8 (e >> 2) + 4 lh
slot = torch.arange(32)
s_, lh_, e_ = slot >> 4, (slot >> 3) & 1, 8 * (slot >> 4) + (slot & 7)
key = (e_ & 3) + 8 * (e_ >> 2) + 4 * lh_
vref = sh(vd).reshape(B, heads, L // 32, 32, 32).transpose(-1, -2)[..., key]       # [B, heads, tiles, d, slot]
for form in [(32, 128, 303), (64, 128, 4)]:
    ops.igemm_force(form[0], form[1], 1, 0, form[2])
    try:
        q, kimg, vtimg = ops.linear_qkv(n, pw, heads, L)
    finally:
        ops.igemm_force(0, 0, 0)
    qs, ks, vs = kimg._aldm_f16
    vt = vtimg.view(torch.float16).double().cpu()
    hi, lo = vt[:, :, :, 0] / vs, vt[:, :, :, 1] / vs
    print(f"form {form}: V^T image err hi+lo {float((hi + lo - vref).abs().max() / vref.abs().max()):.2e}  hi only {float((hi - vref).abs().max() / vref.abs().max()):.2e}"
          f"  max|lo| {float(lo.abs().max()):.2e}", flush=True)
# ---- where is the error?
sc = (sh(qd) @ sh(kd).transpose(-1, -2) / math.sqrt(32)) * 1.4426950408889634     # log2 units [B, heads, L, L]
for form in [(64, 128, 4), (32, 128, 303)]:
    ops.igemm_force(form[0], form[1], 1, 0, form[2])
    try:
        q, kimg, vtimg = ops.linear_qkv(n, pw, heads, L)
    finally:
        ops.igemm_force(0, 0, 0)
    a = ops.attention_presplit(q, kimg, vtimg, heads).double().cpu()
    err = (a - ref).abs().view(B, L, heads, 32).amax(-1)          # [B, L, heads]
    top = torch.topk(err.flatten(), 6)
    print(f"form {form}: rows with the largest error (b, query, head): err, row max score (log2), 2nd max, argmax tile, frac part of max")
    for v, idx in zip(top.values, top.indices):
        b_, r = divmod(int(idx), L * heads)
        qi, h_ = divmod(r, heads)
        row = sc[b_, h_, qi]
        t2 = torch.topk(row, 2)
        print(f"   ({b_}, {qi}, {h_}): {float(v) / float(ref.abs().max()):.2e}  max {float(t2.values[0]):.3f} 2nd {float(t2.values[1]):.3f} argmax key {int(t2.indices[0])}"
              f" (tile {int(t2.indices[0]) // 32})  tile maxima {[round(float(row[32 * j:32 * j + 32].max()), 1) for j in range(L // 32)]}")
for form in [(64, 128, 4), (128, 128, 2), (64, 64, 3), (32, 128, 303)]:
    ops.igemm_force(form[0], form[1], 1, 0, form[2])
    try:
        q, kimg, vtimg = ops.linear_qkv(n, pw, heads, L)
    finally:
        ops.igemm_force(0, 0, 0)
    a = ops.attention_presplit(q, kimg, vtimg, heads).double().cpu()
    a2 = ops.attention_presplit(q, kimg, vtimg, heads).double().cpu()
    err = (a - ref).abs().view(B, L, heads, 32).amax(-1) / ref.abs().max()
    top = torch.topk(err.flatten(), 12)
    rows = []
    for v, idx in zip(top.values, top.indices):
        b_, r = divmod(int(idx), L * heads)
        qi, h_ = divmod(r, heads)
        rows.append(f"({b_},{qi},{h_}):{float(v):.1e}")
    qerr = (q.double().cpu() - qd).abs().view(B, L, heads, 32).amax(-1) / qd.abs().max()
    print(f"form {form}: repeatable {bool(torch.equal(a, a2))}; worst rows {' '.join(rows)}; q err of row (0,78,3) {float(qerr[0, 78, 3]):.2e}")
# ---- emulate the f16x3 attention arithmetic of (batch 0, head 3) in fp64 from the ACTUAL images
ops.igemm_force(64, 128, 1, 0, 4)
try:
    q, kimg, vtimg = ops.linear_qkv(n, pw, heads, L)
finally:
    ops.igemm_force(0, 0, 0)
a = ops.attention_presplit(q, kimg, vtimg, heads).double().cpu()
qs, ks, vs = kimg._aldm_f16
b_, h_ = 0, 3
qm = (q[b_, :, h_ * 32:(h_ + 1) * 32].float().cpu() * torch.tensor(32 ** -0.5 * 1.4426950408889634 * qs, dtype=torch.float32))
q_hi = qm.half(); q_lo = (qm - q_hi.float()).half()
kf = kimg.view(torch.float16).cpu().view(B, L, heads, 2, 32)[b_, :, h_]
k_hi, k_lo = kf[:, 0], kf[:, 1]
S = (q_hi.double() @ k_hi.double().t() + q_hi.double() @ k_lo.double().t() + q_lo.double() @ k_hi.double().t()) / (qs * ks)   # log2 units
S_exact = (sh(qd)[b_, h_] @ sh(kd)[b_, h_].t()) / math.sqrt(32) * 1.4426950408889634
print(f"emulated f16x3 scores vs fp64: max abs err {float((S - S_exact).abs().max()):.2e} log2 units; row 78: {float((S[78] - S_exact[78]).abs().max()):.2e}")
P = torch.exp2(S - S.amax(-1, keepdim=True))
vt = vtimg.view(torch.float16).double().cpu()[b_, h_]                    # [tiles, 2, d, slot]
vv = torch.zeros(L, 32, dtype=torch.float64)
for t_ in range(L // 32):
    vv[32 * t_ + key] = ((vt[t_, 0] + vt[t_, 1]) / vs).t()
O = (P @ vv) / P.sum(-1, keepdim=True)
rr = ref.view(B, L, heads, 32)[b_, :, h_]
ak = a.view(B, L, heads, 32)[b_, :, h_]
print(f"emulation (exact softmax of the emulated scores, exact P.V) vs fp64 ref: {float((O - rr).abs().max() / ref.abs().max()):.2e}; row 78 {float((O[78] - rr[78]).abs().max() / ref.abs().max()):.2e}")
print(f"kernel vs emulation: {float((ak - O).abs().max() / ref.abs().max()):.2e}; row 78 {float((ak[78] - O[78]).abs().max() / ref.abs().max()):.2e}")
print("row 78 kernel - ref:", ((ak[78] - rr[78]) / ref.abs().max()).numpy().round(7)[:8], " top-2 value rows differ by", float((vv[116] - vv[int(torch.topk(S_exact[78], 2).indices[1])]).abs().max()))
# ---- which part is the kernel losing in row 78?  emulate with one low part dropped at a time
def emu(drop):
    qh, ql, kh, kl = q_hi.double(), q_lo.double(), k_hi.double(), k_lo.double()
    S_ = qh @ kh.t()
    if drop != "q_lo": S_ = S_ + ql @ kh.t()
    if drop != "k_lo": S_ = S_ + qh @ kl.t()
    S_ = S_ / (qs * ks)
    m_ = torch.ceil(S_.amax(-1, keepdim=True))
    P_ = torch.exp2(S_ - m_ + 15.0)
    ph = P_.float().half(); pl = (P_.float() - ph.float()).half()
    vh = torch.zeros(L, 32, dtype=torch.float64); vl = torch.zeros(L, 32, dtype=torch.float64)
    for t_ in range(L // 32):
        vh[32 * t_ + key] = vt[t_, 0].t(); vl[32 * t_ + key] = vt[t_, 1].t()
    O_ = ph.double() @ vh
    if drop != "p_lo": O_ = O_ + pl.double() @ vh
    if drop != "v_lo": O_ = O_ + ph.double() @ vl
    return O_ / vs / P_.sum(-1, keepdim=True)
for drop in ("none", "q_lo", "k_lo", "p_lo", "v_lo"):
    O_ = emu(drop)
    print(f"drop {drop:5s}: emulation vs ref row 78 {float((O_[78] - rr[78]).abs().max() / ref.abs().max()):.2e}; kernel vs this emulation row 78 {float((ak[78] - O_[78]).abs().max() / ref.abs().max()):.2e}")
# ---- all rows of all (b, h): kernel vs the full emulation; print the deviating rows with their per-tile integer references
aall = a.view(B, L, heads, 32)
bad = []
for b2 in range(B):
    for h2 in range(heads):
        qm2 = (q[b2, :, h2 * 32:(h2 + 1) * 32].float().cpu() * torch.tensor(32 ** -0.5 * 1.4426950408889634 * qs, dtype=torch.float32))
        qh2 = qm2.half(); ql2 = (qm2 - qh2.float()).half()
        kf2 = kimg.view(torch.float16).cpu().view(B, L, heads, 2, 32)[b2, :, h2]
        S2 = (qh2.double() @ kf2[:, 0].double().t() + qh2.double() @ kf2[:, 1].double().t() + ql2.double() @ kf2[:, 0].double().t()) / (qs * ks)
        vt2 = vtimg.view(torch.float16).double().cpu()[b2, h2]
        vv2 = torch.zeros(L, 32, dtype=torch.float64)
        for t_ in range(L // 32):
            vv2[32 * t_ + key] = ((vt2[t_, 0] + vt2[t_, 1]) / vs).t()
        P2 = torch.exp2(S2 - S2.amax(-1, keepdim=True))
        O2 = (P2 @ vv2) / P2.sum(-1, keepdim=True)
        dev = (aall[b2, :, h2] - O2).abs().amax(-1) / ref.abs().max()
        for r_ in torch.nonzero(dev > 4e-6).flatten().tolist():
            tm = [float(S2[r_, 32 * j:32 * j + 32].max()) for j in range(L // 32)]
            bad.append((b2, r_, h2, float(dev[r_]), [round(x, 3) for x in tm]))
print(f"{len(bad)} rows deviate from the emulation by > 4e-6:")
for x in bad[:12]:
    print("  ", x)
# ---- which key's probability is off in row (0, 78, 3)?  least squares over the 10 largest keys
b2, r2, h2 = 0, 78, 3
Sx = S_exact[r2]
top = torch.topk(Sx, 10).indices
pe = torch.softmax(Sx * math.log(2.0), -1)
resid = (ak[r2] - rr[r2])                       # kernel - ref  (32 dims)
A = (vv[top] - rr[r2]).t()                       # d(out)/d(p_j) = v_j - out   [32, 10]
sol = torch.linalg.lstsq(A, resid.unsqueeze(1)).solution.flatten()
print("row (0,78,3): key, tile, in-tile idx, score, p, fitted dp/p")
for j, kidx in enumerate(top.tolist()):
    print(f"   key {kidx:3d} tile {kidx // 32} idx {kidx % 32:2d}  score {float(Sx[kidx]):8.3f}  p {float(pe[kidx]):.3e}  dp/p {float(sol[j] / pe[kidx]):+.2e}")

"""Measure the best (block tile, split-K) for every igemm geometry a model's sampling job launches and
write audioldm2_amd/tuning-style JSON.  Geometry keys are collected from one short eager job
(ops.TUNE_LOG), then each unique key is re-created with synthetic buffers and every candidate is timed
with events on the launch stream.
Usage (GPU box): python tools/igemm_autotune.py [--mma bf16x6] out.json [model ...]   (default: audioldm2-full)
With --mma bf16x6 the candidates run on the bf16-split kernels and each shape is also timed on the fp32 MFMA with
its fp32-table configuration; the entry records which path won (5th value: 1 = fp32 MFMA)."""
import ctypes as C
import json
import math
import os
import sys

import torch

os.environ["ALDM_NO_GRAPH"] = "1"
os.environ["ALDM_NO_TUNING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm2_amd import lib as L  # noqa: E402
from audioldm2_amd import ops  # noqa: E402
from audioldm2_amd.pipeline import build_model, make_batch_for_text_to_audio, seed_everything  # noqa: E402

BX = False
if "--mma" in sys.argv:
    i = sys.argv.index("--mma")
    BX = sys.argv[i + 1] == "bf16x6"
    del sys.argv[i:i + 2]
TILES = [(128, 128), (64, 128), (128, 64), (64, 64)]
SPLITS = [1, 2, 3, 4, 6, 8, 12, 16]


def collect(model, B):
    torch.manual_seed(1234)
    ld = build_model(model_name=model).cuda()
    batch = make_batch_for_text_to_audio("synthetic prompt", batchsize=B)
    seed_everything(42)
    ld.latent_t_size = 128 if "48k" in model else 256
    ops.TUNE_LOG = []
    ld.generate_batch(batch, unconditional_guidance_scale=3.5, ddim_steps=2, n_gen=1, duration=10)
    keys, ops.TUNE_LOG = ops.TUNE_LOG, None
    from collections import Counter
    # weight = launches per 200-step job: UNet launches repeat 200x, VAE/vocoder once
    return Counter(keys)


def make_desc(key):
    v = [int(t) for t in key.split(",")]
    f = dict(zip(ops._TUNE_FIELDS, v[:-1]))
    pre = v[-1]
    d = L.IgemmDesc()
    for k, val in f.items():
        setattr(d, k, val)
    Cin = f["C1"] + f["C2"]
    K = f["KH"] * f["KW"] * Cin
    d.K = K
    pix1 = f["pix1"] or f["C1"]
    npix = f["B"] * f["H"] * f["W"]
    batch = max(f["batch"], 1)
    keep = []

    def buf(n):
        t = torch.randn(int(n), device="cuda") * 0.5
        keep.append(t)
        return t
    sx = (npix - 1) * pix1 + f["C1"]
    d.stride_x = sx if batch > 1 else 0
    d.x1 = buf(sx * batch).data_ptr()
    if f["C2"]:
        d.x2 = buf(npix * f["C2"]).data_ptr()
    N = f["N"]
    if f["b_mode"] == L.B_PACKED:
        wn = ((K + 3) // 4) * ((N + 31) // 32 * 32) * 4
        d.ldb = 0
    else:
        wn = N * K
        d.ldb = K
    d.stride_w = wn if batch > 1 and f["b_mode"] == L.B_NT else (wn if batch > 1 else 0)
    d.w = buf(wn * batch).data_ptr()
    if BX and f["b_mode"] == L.B_PACKED and batch == 1:
        lib = L.load()
        sp = torch.empty(lib.aldm_split_bytes(K, N) // 4, device="cuda", dtype=torch.int32)
        keep.append(sp)
        L.check(lib.aldm_pack_split_bf16(d.w, sp.data_ptr(), K, N, torch.cuda.current_stream().cuda_stream), "split")
        d.w_split = sp.data_ptr()
    M = f["B"] * f["OH"] * f["OW"]
    geglu = f["epi_mode"] == L.EPI_GEGLU
    ldo = N // 2 if geglu else N
    d.ldo = ldo
    if f["out_mul"]:
        out_len = f["OW"] * f["out_mul"] + 8
        d.out_len = out_len
        d.out_off = 0
        on = f["B"] * out_len * ldo
    else:
        on = M * ldo
    d.stride_o = on if batch > 1 else 0
    d.out = buf(on * batch).data_ptr()
    d.bias = buf(N).data_ptr()
    d.alpha = 1.0
    if pre in (1, 2):
        d.pre_scale = buf(f["B"] * Cin).data_ptr()
        d.pre_shift = buf(f["B"] * Cin).data_ptr()
    d.pre_act = {0: 0, 1: 0, 2: L.ACT_SILU, 3: L.ACT_LRELU, 4: L.ACT_SILU}[pre]
    d.pre_slope = 0.1
    return d, keep, M, N, K


def time_launch(lib, d, reps):
    """us per launch, HIP-graph timed: `reps` launches captured once and replayed (round 3: the eager ctypes launch path costs
    ~17 us per call and hid every kernel shorter than that — most of the transformer blocks' GEMMs)."""
    st = torch.cuda.current_stream().cuda_stream
    ops._workspace(lib, d, torch.device("cuda", torch.cuda.current_device()))
    for _ in range(2):
        rc = lib.aldm_igemm(C.byref(d), st)
        if rc:
            return None
    torch.cuda.synchronize()
    reps = max(reps, 8)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cs = torch.cuda.current_stream().cuda_stream
        for _ in range(reps):
            lib.aldm_igemm(C.byref(d), cs)
    g.replay()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best  # us


def tune(key, lib):
    d, keep, M, N, K = make_desc(key)
    nk = (K + 31) // 32
    flops = 2.0 * M * N * K * max(d.batch, 1)
    reps = 3 if flops > 2e10 else 8
    d.hint_bm = d.hint_bn = d.hint_splits = d.hint_kgroups = d.hint_mma = 0
    t_auto = time_launch(lib, d, reps)
    bm0, bn0, fl, sp0, kg0 = C.c_int(), C.c_int(), C.c_int64(), C.c_int(), C.c_int()
    lib.aldm_igemm_plan(C.byref(d), C.byref(bm0), C.byref(bn0), C.byref(fl), C.byref(sp0), C.byref(kg0), None)
    best = (t_auto, bm0.value, bn0.value, sp0.value, kg0.value)
    geglu = d.epi_mode == L.EPI_GEGLU
    tiles = [(128, 32)] if N <= 32 else TILES
    for bm, bn in tiles:
        if geglu and bn != 128:
            continue
        if bn > 64 and N <= 64 and not geglu:
            continue
        for sp in SPLITS:
            if sp > 1 and (geglu or N % 4 or nk < 8 or nk // sp < 2):
                continue
            blocks = math.ceil(M / bm) * math.ceil(N / bn) * max(d.batch, 1) * sp
            if sp > 1 and blocks > 3072:
                continue
            for kg in ((1, 2) if (bm, bn) == (64, 64) else (1,)):
                if kg == 2 and math.ceil(nk / sp) < 4:
                    continue
                if (bm, bn, sp, kg) == (bm0.value, bn0.value, sp0.value, kg0.value):
                    continue
                d.hint_bm, d.hint_bn, d.hint_splits, d.hint_kgroups = bm, bn, sp, kg
                t = time_launch(lib, d, reps)
                if t is not None and t < best[0]:
                    best = (t, bm, bn, sp, kg)
    best = best + (0,)
    if BX and d.w_split:
        # the same shape on the fp32 MFMA with its own tuned configuration
        h = F32_TABLE.get(key, [0, 0, 0, 0])
        d.hint_bm, d.hint_bn, d.hint_splits, d.hint_kgroups = h[:4]
        d.hint_mma = 1
        t = time_launch(lib, d, reps)
        if t is not None and t < 0.97 * best[0]:
            best = (t, h[0], h[1], h[2], h[3], 1)
    return t_auto, best, (bm0.value, bn0.value, sp0.value, kg0.value), flops


F32_TABLE = {}


def main():
    global F32_TABLE
    out = sys.argv[1]
    if BX:
        ops.set_mma("bf16x6")
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "audioldm2_amd", "tuning",
                            "mi355x_igemm.json")
        with open(path) as f:
            F32_TABLE = json.load(f)["entries"]
    models = sys.argv[2:] or ["audioldm2-full"]
    lib = L.load()
    entries, report = {}, []
    counts = {}
    for m in models:
        c = collect(m, 8)
        for k, n in c.items():
            counts[k] = max(counts.get(k, 0), n)
        print(f"# {m}: {len(c)} unique igemm geometries", flush=True)
    saved = total = 0.0
    for i, (key, n) in enumerate(sorted(counts.items())):
        t_auto, best, auto_cfg, flops = tune(key, lib)
        total += t_auto * n
        if best[0] < 0.97 * t_auto:
            entries[key] = ([best[1], best[2], best[3], best[4]] + ([best[5]] if BX else []) +
                            [round(best[0], 1), round(t_auto, 1)])
            saved += (t_auto - best[0]) * n
        report.append(f"{key} n={n} auto {auto_cfg} {t_auto:.1f}us -> best ({best[1]},{best[2]},{best[3]},{best[4]},mma={best[5]}) {best[0]:.1f}us"
                      f" {flops/best[0]/1e6:.1f} TF")
        print(report[-1], flush=True)
    print(f"# {len(entries)} of {len(counts)} geometries tuned; {saved/1e3:.2f} ms saved of {total/1e3:.2f} ms per 2-step job"
          " (launch counts of the 2-step collection job)")
    with open(out, "w") as f:
        json.dump({"device": torch.cuda.get_device_name(0), "models": models,
                   "fields": list(ops._TUNE_FIELDS) + ["pre_mode"],
                   "mma": "bf16x6" if BX else "f32",
                   "note": ("value = [BM, BN, splits, kgroups, mma (1 = fp32 MFMA), tuned_us, cost_model_us]" if BX else
                            "value = [BM, BN, splits, kgroups, tuned_us, cost_model_us]"), "entries": entries}, f, indent=0)


if __name__ == "__main__":
    main()

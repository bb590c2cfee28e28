"""Measure the best (block tile, LDS ring depth, split-K) of the DMA-fed igemm (csrc/igemm_dma.h) for every geometry a
model's sampling job launches on it and write audioldm2_amd/tuning/mi355x_igemm_dma.json ({key: [BM, BN, splits, stages,
best_us, auto_us]}).  Keys are collected from one short eager job (ops.TUNE_LOG, ",dma" keys).
Usage (GPU box): python tools/dma_autotune.py out.json [model ...]"""
import ctypes as C
import json
import math
import os
import sys

import torch

sys.argv_saved = list(sys.argv)
sys.argv = [sys.argv[0], "--mma", "bf16x6"]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import igemm_autotune as at  # noqa: E402  (sets ALDM_NO_GRAPH / ALDM_NO_TUNING, BX = True)
sys.argv = sys.argv_saved
from audioldm2_amd import lib as L  # noqa: E402
from audioldm2_amd import ops  # noqa: E402

CFGS = {3: [(256, 128, 2), (128, 128, 3), (128, 128, 2), (64, 128, 4), (64, 128, 2), (128, 64, 4), (128, 64, 2), (64, 64, 3),
            (64, 64, 2)],
        2: [(256, 128, 3), (256, 128, 2), (128, 128, 4), (128, 128, 2), (64, 128, 6), (64, 128, 4), (64, 128, 2), (128, 64, 6),
            (128, 64, 4), (128, 64, 2), (64, 64, 6), (64, 64, 3), (64, 64, 2)]}
# loader-wave form (csrc/igemm_dma_lw.h): stages = 200 + ring depth; persistent wave-specialised form (igemm_dma_ws.h): 100 + depth
# (tried without split-K only; a launch it cannot run reports n/a and is skipped).  Both are bitwise igemm_dma_kernel's results.
LW_CFGS = {3: [(128, 128, 202), (128, 128, 203), (64, 128, 202), (64, 128, 204), (128, 64, 202), (128, 64, 204), (64, 64, 202),
               (64, 64, 203)],
           2: [(128, 128, 202), (128, 128, 204), (64, 128, 202), (64, 128, 203), (64, 128, 204), (128, 64, 202), (128, 64, 203),
               (128, 64, 204), (64, 64, 202), (64, 64, 203), (64, 64, 204)]}
WS_CFGS = {3: [(64, 128, 102), (64, 128, 103), (64, 64, 102)],
           2: [(64, 128, 102), (64, 128, 103), (128, 64, 102), (64, 64, 103), (64, 64, 104)]}
# operand-stationary form for short K (csrc/igemm_dma_os.h): "tile" 32x128, stages = 300 + ring depth; K = 256 / 384 1x1 launches only
OS_CFGS = {3: [(32, 128, 302), (32, 128, 303)], 2: [(32, 128, 302), (32, 128, 303), (32, 128, 304)]}
# halo-patch form for 3x3 / stride-1 / pad-1 convolutions (csrc/igemm_dma_halo.h): stages = 400 + 10 * w8 + weight-ring depth
HALO_CFGS = {3: [(256, 128, 402), (128, 128, 402), (128, 128, 403), (128, 128, 404), (128, 128, 412), (128, 128, 413)],
             2: [(256, 128, 402), (256, 128, 403), (128, 128, 403), (128, 128, 404), (128, 128, 413)]}
SPLITS = [1, 2, 3, 4, 6, 8]
PARTS = 2 if os.environ.get("ALDM_MMA") == "bf16x3" else 3
SUFFIX = ",dma2" if PARTS == 2 else ",dma"


def halo_geometry(key):
    f = dict(zip(ops._TUNE_FIELDS, (int(t) for t in key.split(",")[:len(ops._TUNE_FIELDS)])))
    return (f["KH"], f["KW"], f["SH"], f["SW"], f["PH"], f["PW"], f["DH"], f["DW"], f["up_h"], f["up_w"]) == (3, 3, 1, 1, 1, 1, 1, 1, 1, 1) \
        and f["out_mul"] == 0 and f["W"] & (f["W"] - 1) == 0 and (f["H"] * f["W"]) % 128 == 0 and f["C1"] % 32 == 0 and f["C2"] == 0


def tune_halo(key, lib, entry):
    """The geometry's current table entry (or the cost model's choice) against every halo form it can run."""
    d, keep, M, N, K = at.make_desc(key[:-len(SUFFIX)])
    npix = d.B * d.H * d.W
    x = torch.randn(npix, d.C1, device="cuda")
    img = ops.split_rows(x)
    d.a_split = img.data_ptr()
    d.split_parts = PARTS
    if PARTS == 2:
        w2 = torch.empty(lib.aldm_split_bytes_parts(K, N, 2) // 4, device="cuda", dtype=torch.int32)
        keep.append(w2)
        L.check(lib.aldm_pack_split_bf16_parts(d.w, w2.data_ptr(), K, N, 2, torch.cuda.current_stream().cuda_stream), "split2")
        d.w_split = w2.data_ptr()
    d.x1 = None
    d.pre_scale = d.pre_shift = None
    d.pre_act = 0
    flops = 2.0 * M * N * K
    reps = 3 if flops > 2e10 else 8
    d.hint_bm = d.hint_bn = d.hint_splits = d.hint_stages = 0
    if entry:
        d.hint_bm, d.hint_bn, d.hint_splits, d.hint_stages = entry[:4]
    t_cur = at.time_launch(lib, d, reps)
    if t_cur is None:
        return None, None, flops, (M, N, K)
    best = (t_cur, None)
    cpb = d.C1 // 32
    for bm, bn, st in HALO_CFGS[PARTS]:
        for sp in (1, 2, 3, 4, 6):
            if sp > 1 and (cpb % sp or N % 4 or M * N * sp > (1 << 26)):
                continue
            d.hint_bm, d.hint_bn, d.hint_splits, d.hint_stages = bm, bn, sp, st
            if lib.aldm_igemm_plan_stages(C.byref(d)) != st:
                continue
            ops._workspace(lib, d, torch.device("cuda", torch.cuda.current_device()))
            t = at.time_launch(lib, d, reps)
            if t is not None and t < best[0]:
                best = (t, [bm, bn, sp, st])
    return t_cur, best, flops, (M, N, K)


def tune(key, lib):
    d, keep, M, N, K = at.make_desc(key[:-len(SUFFIX)])
    npix = d.B * d.H * d.W
    # operands = split images of random fp32 data (random bit patterns would change the chip's power draw, i.e. its clock)
    x = torch.randn(npix, d.C1, device="cuda")
    img = ops.split_rows(x)
    assert img.parts == PARTS
    d.a_split = img.data_ptr()
    d.split_parts = PARTS
    if PARTS == 2:
        w2 = torch.empty(lib.aldm_split_bytes_parts(K, N, 2) // 4, device="cuda", dtype=torch.int32)
        keep.append(w2)
        L.check(lib.aldm_pack_split_bf16_parts(d.w, w2.data_ptr(), K, N, 2, torch.cuda.current_stream().cuda_stream), "split2")
        d.w_split = w2.data_ptr()
    d.x1 = None
    d.pre_scale = d.pre_shift = None
    d.pre_act = 0
    nk = K // 32
    flops = 2.0 * M * N * K
    reps = 3 if flops > 2e10 else 8
    d.hint_bm = d.hint_bn = d.hint_splits = d.hint_stages = 0
    t_auto = at.time_launch(lib, d, reps)
    if t_auto is None:
        return None, (None, 0, 0, 0, 0), flops, (M, N, K)
    best = (t_auto, 0, 0, 0, 0)
    geglu = d.epi_mode == L.EPI_GEGLU
    for bm, bn, st in CFGS[PARTS]:
        if geglu and bn != 128:
            continue
        if bn > 64 and N <= 64 and not geglu:
            continue
        if bm > 64 and M <= 64:
            continue
        for sp in SPLITS:
            if sp > 1 and (geglu or N % 4 or nk < 8 or nk // sp < 2):
                continue
            blocks = math.ceil(M / bm) * math.ceil(N / bn) * sp
            if sp > 1 and blocks > 2048:
                continue
            d.hint_bm, d.hint_bn, d.hint_splits, d.hint_stages = bm, bn, sp, st
            t = at.time_launch(lib, d, reps)
            if t is not None and t < best[0]:
                best = (t, bm, bn, sp, st)
    for bm, bn, st in LW_CFGS[PARTS] + WS_CFGS[PARTS] + (OS_CFGS[PARTS] if K in (256, 384) else []):
        if geglu and bn != 128:
            continue
        if bn > 64 and N <= 64 and not geglu:
            continue
        if bm > 64 and M <= 64:
            continue
        for sp in (SPLITS if 200 <= st < 300 else [1]):
            if sp > 1 and (geglu or N % 4 or nk < 8 or nk // sp < 2):
                continue
            blocks = math.ceil(M / bm) * math.ceil(N / bn) * sp
            if sp > 1 and blocks > 2048:
                continue
            d.hint_bm, d.hint_bn, d.hint_splits, d.hint_stages = bm, bn, sp, st
            t = at.time_launch(lib, d, reps)
            if t is not None and lib.aldm_igemm_plan_stages(C.byref(d)) != st:
                continue   # a hinted launch the kernel cannot run silently stays on igemm_dma_kernel: not this candidate
            if t is not None and t < best[0]:
                best = (t, bm, bn, sp, st)
    return t_auto, best, flops, (M, N, K)


def main():
    out = sys.argv[1]
    models = sys.argv[2:] or ["audioldm2-full"]
    ops.set_mma("bf16x3" if PARTS == 2 else "bf16x6")
    ops.set_dma(True)
    lib = L.load()
    counts = {}
    for m in models:
        c = at.collect(m, 8)
        for k, n in c.items():
            if k.endswith(SUFFIX):
                counts[k] = max(counts.get(k, 0), n)
        print(f"# {m}: {sum(1 for k in c if k.endswith(SUFFIX))} unique DMA-fed igemm geometries ({PARTS} parts)", flush=True)
    entries, total, saved = {}, 0.0, 0.0
    # $DMA_TUNE_MERGE=<table.json>: start from that table's entries and tune only geometries it does not hold;
    # $DMA_TUNE_MIN_COUNT=n: skip geometries launched fewer than n times in the 2-step job (VAE / vocoder launches run once)
    merge = os.environ.get("DMA_TUNE_MERGE")
    if merge and os.path.exists(merge):
        with open(merge) as f:
            entries = dict(json.load(f)["entries"])
        print(f"# merging into {merge}: {len(entries)} existing entries", flush=True)
    min_count = int(os.environ.get("DMA_TUNE_MIN_COUNT", "0"))
    known = set(entries)
    only_os = os.environ.get("DMA_TUNE_ONLY_OS", "0") == "1"   # re-tune (only) the geometries the operand-stationary kernel can run
    only_halo = os.environ.get("DMA_TUNE_ONLY_HALO", "0") == "1"   # ... the halo-patch kernel can run, against their current entries

    def os_geometry(key):
        f = key.split(",")
        return f[8] == "1" and f[9] == "1" and f[3] in ("256", "384") and f[4] == "0"
    for key, n in sorted(counts.items()):
        if only_halo:
            if not halo_geometry(key) or n < min_count:
                continue
            t_cur, best, flops, mnk = tune_halo(key, lib, entries.get(key))
            if t_cur is None:
                continue
            total += t_cur * n
            line = f"M{mnk[0]} N{mnk[1]} K{mnk[2]} n={n} current {entries.get(key, ['auto'])[:4]} {t_cur:.1f}us"
            if best[1] and best[0] < 0.98 * t_cur:
                auto_us = entries[key][5] if key in entries and len(entries[key]) > 5 else round(t_cur, 1)
                entries[key] = best[1] + [round(best[0], 1), auto_us]
                saved += (t_cur - best[0]) * n
                line += f" -> halo {best[1]} {best[0]:.1f}us {flops / best[0] / 1e6:.1f} TF ({100 * (best[0] / t_cur - 1):+.0f} %)"
            else:
                line += " -> kept"
            print(line, flush=True)
            continue
        if only_os:
            if not os_geometry(key) or n < min_count:
                continue
        elif key in known or n < min_count:
            continue
        t_auto, best, flops, mnk = tune(key, lib)
        if t_auto is None:
            print(f"# {key}: the default configuration does not launch on synthetic buffers, skipped", flush=True)
            continue
        total += t_auto * n
        if best[1] and best[0] < 0.97 * t_auto:
            entries[key] = [best[1], best[2], best[3], best[4], round(best[0], 1), round(t_auto, 1)]
            saved += (t_auto - best[0]) * n
        elif only_os and key in entries:
            del entries[key]   # the cost model's own choice is (now) within 3 % of the best
        print(f"M{mnk[0]} N{mnk[1]} K{mnk[2]} n={n} auto {t_auto:.1f}us -> best {best[1]}x{best[2]} k{best[3]} st{best[4]} "
              f"{best[0]:.1f}us {flops / best[0] / 1e6:.1f} TF", flush=True)
    with open(out, "w") as f:
        json.dump({"device": "MI355X", "kernel": "igemm_dma_kernel", "parts": PARTS, "entries": entries}, f, indent=0, sort_keys=True)
    print(f"# {len(entries)} tuned entries, {saved / 1e3:.2f} ms of {total / 1e3:.2f} ms (2-step job) saved", flush=True)


if __name__ == "__main__":
    main()

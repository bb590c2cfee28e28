"""Tune the DMA-fed igemm geometry table INSIDE the replayed UNet step (round 6).  tools/dma_autotune.py times every geometry alone, in a
graph of back-to-back launches: warm weights, a chip at its power cap — and the step is neither (the re-tune of the 3-part table that way
made every configuration's step slower, profiles/r06_step_ab_tables_3part_rejected.txt).  This tool changes ONE geometry's entry at a
time in the live table, re-captures the step graph and measures the step itself (bench.unet_step_probe), keeping a candidate only when the
step gets faster by more than the noise twice in a row.  Coordinate descent over the geometries with the most launches x time.
Usage (GPU box): ALDM_MMA=bf16x6|f16x3|bf16x3 python tools/instep_autotune.py out.json [model] [n_keys]"""
import json
import os
import sys
import time
from collections import Counter

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from audioldm2_amd import ops  # noqa: E402
from audioldm2_amd.pipeline import build_model, make_batch_for_text_to_audio  # noqa: E402

out_path = sys.argv[1]
model = sys.argv[2] if len(sys.argv) > 2 else "audioldm2-full"
n_keys = int(sys.argv[3]) if len(sys.argv) > 3 else 40
MARGIN = float(os.environ.get("INSTEP_MARGIN_MS", "0"))   # 0: 0.06 % of the step, at least 0.012 ms (set after the first measurement)
BUDGET_S = float(os.environ.get("INSTEP_BUDGET_S", "1500"))
mode = ops.MMA_MODE
B = 8
torch.manual_seed(1234)
ld = build_model(model_name=model).cuda()
ld.latent_t_size = 256 if "48k" not in model else 128
batch = make_batch_for_text_to_audio("synthetic prompt", batchsize=B)
unet = ld.model.diffusion_model


def step_ms(n=1):
    unet.drop_step_caches()
    return min(bench.unet_step_probe(ld, batch, B) for _ in range(n))


# which geometries the step launches, and how often (one eager pass with the key log on)
t_first = step_ms()
ops.TUNE_LOG = []
cond = ld.get_learned_conditioning_dict(batch)
uncond = {k: ld.cond_stage_models[m["model_idx"]].get_unconditional_condition(B) for k, m in ld.cond_stage_model_metadata.items()}
x = torch.randn(B, ld.channels, ld.latent_t_size, ld.latent_f_size).cuda()
ld.apply_model_cfg(x, torch.full((2 * B,), 501.0).cuda(), cond, uncond)
torch.cuda.synchronize()
ONLY = os.environ.get("INSTEP_ONLY", "")      # "dma" / "dma2": tune only the 3-part / 2-part geometries of the step (f16x3 launches both)
counts, ops.TUNE_LOG = Counter(k for k in ops.TUNE_LOG if (k.endswith(",dma") or k.endswith(",dma2")) and
                               (not ONLY or k.endswith("," + ONLY))), None
tabs = {"dma": ops._tuned_table("dma"), "dma2": ops._tuned_table("dma2")}


def fields(key):
    v = key.split(",")
    return dict(zip(ops._TUNE_FIELDS, (int(t) for t in v[:len(ops._TUNE_FIELDS)]))), v[-1]


def weight(key):
    f, _ = fields(key)
    M = f["B"] * f["OH"] * f["OW"]
    K = f["KH"] * f["KW"] * (f["C1"] + f["C2"])
    return counts[key] * (2.0 * M * f["N"] * K / 2.0e8 + 6.0)     # launches x (a 200 TFLOP/s body + a 6 us floor), in us


# the forms tools/dma_autotune.py tries (tile, ring depth / form code), per image format
CFGS = {3: [(256, 128, 2), (128, 128, 3), (128, 128, 2), (64, 128, 4), (64, 128, 2), (128, 64, 4), (128, 64, 2), (64, 64, 3), (64, 64, 2)],
        2: [(256, 128, 3), (256, 128, 2), (128, 128, 4), (128, 128, 2), (64, 128, 6), (64, 128, 4), (64, 128, 2), (128, 64, 6),
            (128, 64, 4), (128, 64, 2), (64, 64, 6), (64, 64, 3), (64, 64, 2)]}
LW_CFGS = {3: [(128, 128, 202), (128, 128, 203), (64, 128, 202), (64, 128, 204), (128, 64, 202), (128, 64, 204), (64, 64, 202), (64, 64, 203)],
           2: [(128, 128, 202), (128, 128, 204), (64, 128, 202), (64, 128, 203), (64, 128, 204), (128, 64, 202), (128, 64, 203),
               (128, 64, 204), (64, 64, 202), (64, 64, 203), (64, 64, 204)]}
OS_CFGS = {3: [(32, 128, 302), (32, 128, 303)], 2: [(32, 128, 302), (32, 128, 303), (32, 128, 304)]}
HALO_CFGS = {3: [(256, 128, 402), (128, 128, 402), (128, 128, 403), (128, 128, 404), (128, 128, 412), (128, 128, 413)],
             2: [(256, 128, 402), (256, 128, 403), (128, 128, 403), (128, 128, 404), (128, 128, 413)]}
SPLIT_FORMS = {3: [(64, 64, 203), (128, 64, 204), (64, 128, 204), (128, 128, 203), (256, 128, 2)],
               2: [(64, 64, 203), (128, 64, 204), (64, 128, 204), (128, 128, 204), (256, 128, 2)]}


def candidates(key):
    f, suffix = fields(key)
    parts = 2 if suffix == "dma2" else 3
    M = f["B"] * f["OH"] * f["OW"]
    K = f["KH"] * f["KW"] * (f["C1"] + f["C2"])
    N = f["N"]
    geglu = f["epi_mode"] == 1
    conv3 = (f["KH"], f["KW"], f["SH"], f["SW"], f["PH"], f["PW"], f["DH"], f["DW"], f["up_h"], f["up_w"]) == (3, 3, 1, 1, 1, 1, 1, 1, 1, 1)
    one = f["KH"] == 1 and f["KW"] == 1
    out = []
    for bm, bn, st in CFGS[parts] + LW_CFGS[parts]:
        if (geglu and bn != 128) or (bn > 64 and N <= 64 and not geglu) or (bm > 64 and M <= 64):
            continue
        out.append([bm, bn, 1, st])
    if K // 32 >= 16 and not geglu and N % 4 == 0 and M <= 16384:
        for sp in (2, 3, 4):
            for bm, bn, st in SPLIT_FORMS[parts]:
                if bm <= M and (K // 32) // sp >= 2 and -(-M // bm) * -(-N // bn) * sp <= 2048:
                    out.append([bm, bn, sp, st])
    if one and K in (256, 384) and f["C2"] == 0:
        out += [[bm, bn, 1, st] for bm, bn, st in OS_CFGS[parts]]
    if conv3 and f["C2"] == 0 and f["C1"] % 32 == 0 and f["out_mul"] == 0 and f["W"] & (f["W"] - 1) == 0 and (f["H"] * f["W"]) % 128 == 0:
        out += [[bm, bn, 1, st] for bm, bn, st in HALO_CFGS[parts]]
    return out


keys = sorted(counts, key=weight, reverse=True)[:n_keys]
base = step_ms(3)
if MARGIN <= 0:   # a 45 ms step drifts by more than 0.012 ms over a descent (the large configuration's "wins" did not survive the A/B)
    MARGIN = max(0.012, 0.0006 * base)
print(f"# {model} {mode}: {len(counts)} DMA-fed geometries in a pass, tuning the {len(keys)} heaviest in the step; step {base:.3f} ms "
      f"(first capture {t_first:.3f}); margin {MARGIN} ms", flush=True)
t_start = time.time()
changed = {}
for key in keys:
    if time.time() - t_start > BUDGET_S:
        print("# time budget reached", flush=True)
        break
    tab = tabs["dma2" if key.endswith(",dma2") else "dma"]
    cur = tab.get(key)
    best_cfg, best_t = cur, base
    f, _ = fields(key)
    tried = 0
    for cfg in candidates(key):
        if cur is not None and cfg == list(cur[:4]):
            continue
        tab[key] = cfg
        try:
            t = step_ms()
            tried += 1
            if t < best_t - MARGIN:
                t2 = step_ms()               # confirm: a candidate must win twice
                if t2 < best_t - MARGIN:
                    best_cfg, best_t = cfg, max(t, t2)
        except RuntimeError as e:           # a hint the launch cannot run is dropped by the planner; anything else is a bug worth seeing
            print(f"#   {cfg}: {str(e)[:100]}", flush=True)
    if best_cfg is None:
        tab.pop(key, None)
    else:
        tab[key] = best_cfg
    M = f["B"] * f["OH"] * f["OW"]
    line = f"n={counts[key]:3d} M{M} N{f['N']} K{f['KH'] * f['KW'] * (f['C1'] + f['C2'])} taps{f['KH'] * f['KW']} epi{f['epi_mode']}: " \
           f"{list(cur[:4]) if cur else 'auto'} -> "
    if best_cfg is not cur:
        changed[key] = best_cfg
        line += f"{best_cfg}  step {base:.3f} -> {best_t:.3f} ms"
        base = step_ms(2)                    # re-base on a fresh measurement of the accepted table
        line += f" (re-based {base:.3f})"
    else:
        line += f"kept ({tried} candidates)"
    print(line, flush=True)
final = step_ms(3)
print(f"# {len(changed)} entries changed; step now {final:.3f} ms; {time.time() - t_start:.0f} s", flush=True)
with open(out_path, "w") as fh:
    json.dump({"mode": mode, "model": model, "changed": changed, "step_ms": final}, fh, indent=0, sort_keys=True)

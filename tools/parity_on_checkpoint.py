#!/usr/bin/env python
"""parity_on_checkpoint.py — the one-command parity report on TRAINED weights.  CHECKER, not product: it
imports `oracle/` (test infrastructure) to run the REAL reference.

Every parity number in this repository was measured on deterministic random-init weights, because no checkpoint is reachable
offline.  Whoever has one (`audioldm2-full.pth`, ... — the files `hf_hub_download` fetches in the reference's
pipeline.py:159-164) closes the claim with:

    python tools/parity_on_checkpoint.py --ckpt audioldm2-full.pth --model audioldm2-full --reference-root /path/to/AudioLDM2

which
  1. [reference stage, CPU] builds the reference's own `LatentDiffusion` (audioldm2/latent_diffusion/models/ddpm.py, imported
     where it lies through oracle/refimport.py) with the synthetic conditioners the fixtures use (the real ones need Hub
     tokenizers; the hot path under test starts at the conditioning tensors), loads `checkpoint["state_dict"]`
     like pipeline.py:172-174 does — every hot-path tensor (`model.diffusion_model.*`, `first_stage_model.*`, the schedule
     buffers, `scale_factor`) must be present — and runs `generate_batch` (ddpm.py:1477) for each `--steps` entry: seed 42,
     CFG 3.5, eta 1.0, one candidate per prompt; latent, mel and waveform go to `--cache` (an .npz);
  2. [hip stage, MI355X] builds this repository's `LatentDiffusion` through `build_model(ckpt_path=...)` (the reference's own
     entry point signature, pipeline.py:142), runs the same jobs on the GPU in the library's default product mode (or --mma)
     and prints, per job: latent / mel relative rms error, waveform rms error, the waveform's own rms and the rms distance
     between two unrelated samples of the batch — next to the bars tests/tolerances.py holds the random-init fixtures to.
`--stage reference` / `--stage hip` run the halves on different machines (the cache travels); `--stage all` runs both.
Exit code 0 = every job inside the bars, 1 = a bar missed, 2 = a stage could not run (no GPU, no reference checkout).

tests/test_host_logic.py::test_parity_on_checkpoint_script_* exercise the script on a random-init checkpoint written in the
reference's format (CPU: the reference stage end to end + `build_model(ckpt_path=...)` loading the same tensors; the hip stage
must refuse without a GPU)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

SEED = 42
HOT_PREFIXES = ("model.diffusion_model.", "first_stage_model.")


def rms(a):
    a = np.asarray(a, dtype=np.float64)
    return float(np.sqrt((a ** 2).mean()))


def seed_all():
    """pipeline.py:20-31 seed_everything(42)."""
    import random
    random.seed(SEED)
    np.random.seed(SEED)
    torch.manual_seed(SEED)


def make_batch(model, B):
    from oracle import cases
    b = cases.e2e_batch(B)
    if "48k" in model:
        b["log_mel_spec"] = torch.zeros((B, 1024, 256))
        b["fbank"] = b["log_mel_spec"]
    return b


def load_state_dict_file(path):
    ckpt = torch.load(path, map_location="cpu")
    sd = ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt
    assert isinstance(sd, dict) and sd, f"{path}: no state_dict"
    return sd


def make_random_checkpoint(path, model):
    """A random-init checkpoint in the reference's format WITHOUT the reference: this repository's module tree carries the reference's
    state-dict keys and shapes (tests/test_host_logic.py pins them to the real classes), so its state dict with deterministic hot-path
    tensors (oracle/weights.py, the generator behind every fixture) is a file both stages load.  For exercising the two stages on machines
    that have no released checkpoint (the CPU suite; tools/gpu/r5_run9.sh runs the hip stage on it)."""
    from audioldm2_amd.pipeline import build_model
    from oracle import cases, weights
    sd = build_model(model_name=model).state_dict()
    hot = {k: tuple(v.shape) for k, v in sd.items() if k.startswith(HOT_PREFIXES)}
    sd.update(weights.make_state_dict(hot, seed=3))
    sd["scale_factor"] = torch.tensor(cases.SCALE_FACTOR)
    sd = {k: v for k, v in sd.items() if not k.startswith(("cond_stage_models.", "clap."))}
    torch.save({"state_dict": sd, "note": "random-init (oracle.weights seed 3), reference key layout"}, path)
    print(f"wrote {path}: {len(sd)} tensors ({len(hot)} hot-path)", flush=True)


def reference_stage(args, jobs):
    """The real reference on the CPU."""
    if args.reference_root:
        os.environ["ALDM_REFERENCE_ROOT"] = args.reference_root
    import importlib
    from oracle import refimport
    importlib.reload(refimport) if args.reference_root else None
    if not refimport.available():
        print(f"parity_on_checkpoint: no reference checkout under {refimport.REF_ROOT} (--reference-root)", file=sys.stderr)
        return None
    refimport.install()
    import audioldm2.utils as ru
    from audioldm2.latent_diffusion.models.ddpm import LatentDiffusion
    from audioldm2_amd.pipeline import default_audioldm_config
    P = ru.default_audioldm_config(args.model)["model"]["params"]
    cond = default_audioldm_config(args.model)["model"]["params"]["cond_stage_config"]   # the fixtures' synthetic conditioners under
    for k in cond:                                                                       # the reference's own cond keys
        cond[k]["params"]["device"] = "cpu"
    P["cond_stage_config"] = cond
    P["device"] = "cpu"
    torch.manual_seed(0)
    ld = LatentDiffusion(**P).eval()
    sd = load_state_dict_file(args.ckpt)
    mine = ld.state_dict()
    hot_missing = [k for k in mine if k.startswith(HOT_PREFIXES) and k not in sd]
    assert not hot_missing, f"{args.ckpt} lacks {len(hot_missing)} hot-path tensors of {args.model}, e.g. {hot_missing[:4]}"
    take = {k: v for k, v in sd.items() if k in mine and not k.startswith(("cond_stage_models.", "clap."))}
    ld.load_state_dict(take, strict=False)   # the conditioners are the synthetic ones: their entries (and EMA copies) stay out
    print(f"[reference] {args.model}: loaded {len(take)} tensors of {len(sd)} in {args.ckpt} "
          f"({sum(k.startswith(HOT_PREFIXES) for k in take)} hot-path); threads {torch.get_num_threads()}", flush=True)
    ld.latent_t_size = 128 if "48k" in args.model else 256
    out = {"model": np.array(args.model), "ckpt": np.array(os.path.basename(args.ckpt))}
    for steps, B in jobs:
        rec = {}
        orig = ld.decode_first_stage

        def hook(z, rec=rec, orig=orig):
            rec["latent"] = z.clone()
            rec["mel"] = orig(z)
            return rec["mel"]
        ld.decode_first_stage = hook
        seed_all()
        t0 = time.time()
        with torch.no_grad():
            wav = ld.generate_batch(make_batch(args.model, B), unconditional_guidance_scale=3.5, ddim_steps=steps, n_gen=1, duration=10)
        ld.decode_first_stage = orig
        tag = f"s{steps}_b{B}"
        out[tag + "_latent"] = rec["latent"].numpy()
        out[tag + "_mel"] = rec["mel"].numpy()
        out[tag + "_wave"] = np.asarray(wav)
        print(f"[reference] {steps} steps, batch {B}: {time.time() - t0:.1f} s on the CPU; wave {tuple(wav.shape)} rms {rms(wav):.4f}",
              flush=True)
    np.savez(args.cache, **out)
    print(f"[reference] wrote {args.cache}", flush=True)
    return out


def hip_stage(args, jobs, ref):
    if not torch.cuda.is_available():
        print("parity_on_checkpoint: the hip stage needs the MI355X (there is no CPU path); run `--stage reference` here and "
              "`--stage hip --cache ...` on the GPU box", file=sys.stderr)
        return None
    from audioldm2_amd import ops
    from audioldm2_amd.pipeline import build_model
    import tolerances as tol
    if args.mma:
        ops.set_mma(args.mma)
    mode = ops.MMA_MODE
    ld = build_model(ckpt_path=args.ckpt, model_name=args.model).cuda()      # pipeline.py:142-179: strict on the hot path
    ld.latent_t_size = 128 if "48k" in args.model else 256
    ok = True
    rows = []
    for steps, B in jobs:
        tag = f"s{steps}_b{B}"
        if tag + "_wave" not in ref:
            print(f"[hip] {tag}: not in {args.cache}, skipped")
            continue
        rec = {}
        orig = ld.decode_first_stage_cl

        def hook(z, rec=rec, orig=orig):
            rec["latent"] = z.clone()
            rec["mel"] = orig(z)
            return rec["mel"]
        ld.decode_first_stage_cl = hook
        seed_all()
        wav = ld.generate_batch(make_batch(args.model, B), unconditional_guidance_scale=3.5, ddim_steps=steps, n_gen=1, duration=10)
        ld.decode_first_stage_cl = orig
        g_lat, g_mel, g_wav = ref[tag + "_latent"], ref[tag + "_mel"], ref[tag + "_wave"]
        lat = rec["latent"].double().cpu().numpy().reshape(g_lat.shape)
        mel = rec["mel"].double().cpu().numpy().reshape(g_mel.shape)
        e_lat, e_mel = rms(lat - g_lat) / rms(g_lat), rms(mel - g_mel) / rms(g_mel)
        e_wav = rms(wav.astype(np.float64) - g_wav)
        between = rms(g_wav[0].astype(np.float64) - g_wav[1].astype(np.float64)) if B >= 2 else float("nan")
        bars = (tol.latent_tol(steps, mode), tol.mel_tol(steps, mode), 1e-3)
        good = e_lat < bars[0] and e_mel < bars[1] and e_wav < bars[2] and (B < 2 or e_wav < 1e-3 * between)
        ok &= good
        rows.append({"steps": steps, "batch": B, "mode": mode, "latent_rel_rms": e_lat, "mel_rel_rms": e_mel, "wave_rms_err": e_wav,
                     "wave_rms": rms(g_wav), "wave_between_samples_rms": between, "bars": bars, "ok": bool(good)})
        print(f"[hip] {args.model} [{mode}] {steps} steps, batch {B}: latent rel rms {e_lat:.2e} (bar {bars[0]:.0e})  mel rel rms "
              f"{e_mel:.2e} (bar {bars[1]:.0e})  wave rms err {e_wav:.3e} (north_star 1e-3; wave rms {rms(g_wav):.3e}, between two "
              f"samples {between:.3e})  {'OK' if good else 'MISSED'}", flush=True)
    print(json.dumps({"parity_on_checkpoint": os.path.basename(args.ckpt), "model": args.model, "jobs": rows, "ok": bool(ok)}))
    return ok


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--ckpt", required=True, help="reference-format checkpoint: torch.save({'state_dict': ...})")
    ap.add_argument("--model", default="audioldm2-full")
    ap.add_argument("--reference-root", default=None, help="checkout of haoheliu/AudioLDM2 (default: $ALDM_REFERENCE_ROOT or /root/reference)")
    ap.add_argument("--steps", default="5,200", help="DDIM step counts, comma separated")
    ap.add_argument("--batch", default="2,1", help="batch per entry of --steps (a single value applies to all)")
    ap.add_argument("--stage", choices=["all", "reference", "hip"], default="all")
    ap.add_argument("--cache", default="parity_reference.npz", help="reference outputs (written by the reference stage, read by the hip stage)")
    ap.add_argument("--mma", choices=["bf16x6", "bf16x3", "f32"], default=None)
    ap.add_argument("--threads", type=int, default=0, help="CPU threads of the reference stage (0: torch default)")
    ap.add_argument("--make-random-ckpt", action="store_true",
                    help="first write --ckpt as a deterministic random-init checkpoint in the reference's key layout (no released checkpoint at hand)")
    args = ap.parse_args()
    if args.make_random_ckpt:
        make_random_checkpoint(args.ckpt, args.model)
    steps = [int(s) for s in args.steps.split(",")]
    bs = [int(b) for b in args.batch.split(",")]
    bs = bs * len(steps) if len(bs) == 1 else bs
    assert len(bs) == len(steps), "--batch needs one value, or one per --steps entry"
    jobs = list(zip(steps, bs))
    if args.threads:
        torch.set_num_threads(args.threads)
    ref = None
    if args.stage in ("all", "reference"):
        ref = reference_stage(args, jobs)
        if ref is None:
            return 2
    if args.stage in ("all", "hip"):
        if ref is None:
            if not os.path.exists(args.cache):
                print(f"parity_on_checkpoint: {args.cache} not found — run `--stage reference` first", file=sys.stderr)
                return 2
            ref = dict(np.load(args.cache))
        ok = hip_stage(args, jobs, ref)
        if ok is None:
            return 2
        return 0 if ok else 1
    return 0


if __name__ == "__main__":
    sys.exit(main())

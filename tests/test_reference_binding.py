"""The reference-side binding of INTEGRATION.md §2, EXECUTED (VERDICT r1 "what's weak" #1):

 (a) CPU, build container only (skipped where /root/reference is absent): the REAL
     `audioldm2.latent_diffusion.models.ddpm.LatentDiffusion` (ddpm.py:640) is constructed through the reference's own
     `instantiate_from_config` (utils.py:95-114) with the two `target` strings of INTEGRATION.md swapped in and
     `ddpm.DDIMSampler` rebound; a reference-shaped state dict strict-loads; the module tree is ours.
 (b) GPU: `audioldm2_amd.ddim.DDIMSampler` driven by a model that exposes ONLY what the reference's LatentDiffusion
     offers a sampler (`apply_model`, `num_timesteps`, `alphas_cumprod`; no `apply_model_cfg` / `prepare_cfg`), i.e. the
     two-sequential-pass CFG fallback the reference's own class would hit (ddim.py:293-296), against the reference
     fixture and against the batched path."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cases, refimport, weights

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rms(a):
    return float(np.sqrt(np.mean(np.square(np.asarray(a, dtype=np.float64)))))


@pytest.mark.skipif(not refimport.available(), reason="needs the reference checkout (build container only)")
def test_real_reference_latent_diffusion_builds_on_our_targets_and_strict_loads():
    refimport.install()
    import audioldm2.latent_diffusion.models.ddpm as rddpm
    import audioldm2.utils as ru

    import audioldm2_amd.ddim as addim
    import audioldm2_amd.hifigan as ahifigan
    import audioldm2_amd.unet as aunet
    import audioldm2_amd.vae as avae
    P = ru.default_audioldm_config("audioldm2-full")["model"]["params"]
    assert P["unet_config"]["target"].endswith("openaimodel.UNetModel")          # utils.py:329
    assert P["first_stage_config"]["target"].endswith("autoencoder.AutoencoderKL")  # utils.py:275
    P["unet_config"]["target"] = "audioldm2_amd.unet.UNetModel"                  # INTEGRATION.md §2, string 1
    P["first_stage_config"]["target"] = "audioldm2_amd.vae.AutoencoderKL"        # string 2
    P["cond_stage_config"] = cases.e2e_cond_config("cpu")
    P["device"] = "cpu"
    saved = rddpm.DDIMSampler
    rddpm.DDIMSampler = addim.DDIMSampler                                        # the rebinding (ddpm.py:19, :1437)
    try:
        torch.manual_seed(0)
        ld = rddpm.LatentDiffusion(**P).eval()
        # the module tree is ours, reached through the reference's own plugin seam
        assert type(ld).__module__ == "audioldm2.latent_diffusion.models.ddpm"
        assert isinstance(ld.model.diffusion_model, aunet.UNetModel)
        assert isinstance(ld.first_stage_model, avae.AutoencoderKL)
        assert isinstance(ld.first_stage_model.vocoder, ahifigan.Generator)
        # state-dict compatibility: exactly the reference's hot-path keys and shapes, strict load of a checkpoint-shaped
        # dict (pipeline.py:172-174 loads checkpoint["state_dict"] this way)
        with open(os.path.join(GOLD, "e2e_statedict_keys.json")) as f:
            ref_shapes = {k: tuple(v) for k, v in json.load(f).items()}
        sd = ld.state_dict()
        hot = {k: tuple(v.shape) for k, v in sd.items()
               if k.startswith("model.diffusion_model.") or k.startswith("first_stage_model.")}
        assert hot == ref_shapes
        new = weights.make_state_dict(ref_shapes, seed=0)
        full = dict(sd)
        full.update(new)
        ld.load_state_dict(full, strict=True)
        w = ld.model.diffusion_model.input_blocks[1][0].in_layers[2].weight
        assert torch.equal(w, new["model.diffusion_model.input_blocks.1.0.in_layers.2.weight"])
        # sample_log builds the sampler by NAME from ddpm's globals (ddpm.py:1437): it is ours now, and it accepts the
        # reference model (which has no apply_model_cfg -> the two-pass CFG fallback, covered on the GPU below)
        s = rddpm.DDIMSampler(ld)
        assert isinstance(s, addim.DDIMSampler) and not hasattr(ld, "apply_model_cfg")
        s.make_schedule(ddim_num_steps=200, ddim_eta=1.0, verbose=False)
        assert s.ddim_timesteps[0] == 1 and s.ddim_timesteps[-1] == 996
        # there is no CPU fallback behind the plugin: the forward refuses a CPU tensor loudly
        with pytest.raises(RuntimeError):
            ld.model.diffusion_model(torch.zeros(1, 8, 256, 16), torch.zeros(1), context_list=[], context_attn_mask_list=[])
    finally:
        rddpm.DDIMSampler = saved


class _ReferenceShapedModel:
    """What ddim.py touches on its model (`:19,42,46-51,188,228,285-296,302,338`): nothing else is exposed."""

    def __init__(self, ld):
        self._ld = ld
        self.num_timesteps = ld.num_timesteps
        self.alphas_cumprod = ld.alphas_cumprod
        self.device = ld.device

    def apply_model(self, x_noisy, t, cond):
        return self._ld.apply_model(x_noisy, t, cond)


@pytest.mark.gpu
def test_ddim_sampler_two_pass_cfg_fallback_matches_reference_fixture_and_batched_path():
    from audioldm2_amd.ddim import DDIMSampler
    from audioldm2_amd.pipeline import build_model, seed_everything
    g = np.load(os.path.join(GOLD, "e2e_full_5step_b2.npz"))
    ld = build_model(model_name="audioldm2-full")
    with open(os.path.join(GOLD, "e2e_statedict_keys.json")) as f:
        shapes = {k: tuple(v) for k, v in json.load(f).items()}
    sd = weights.make_state_dict(shapes, seed=0)
    sd["scale_factor"] = torch.tensor(cases.SCALE_FACTOR)
    ld.load_state_dict(sd, strict=False)
    ld = ld.cuda()
    ld.latent_t_size = 256
    B, steps = 2, 5
    batch = cases.e2e_batch(B)

    def run(model):
        seed_everything(cases.E2E_SEED)
        torch.randn((B, 8, 256, 16))  # RNG contract R1: the posterior draw of the zero-mel encode (ddpm.py:1500)
        cond = ld.get_learned_conditioning_dict(batch)
        uncond = {k: ld.cond_stage_models[m["model_idx"]].get_unconditional_condition(B)
                  for k, m in ld.cond_stage_model_metadata.items()}
        s = DDIMSampler(model)
        z, _ = s.sample(steps, B, (8, 256, 16), cond, eta=1.0, verbose=False, unconditional_guidance_scale=3.5,
                        unconditional_conditioning=uncond)
        return z
    shim = _ReferenceShapedModel(ld)
    assert not hasattr(shim, "apply_model_cfg") and not hasattr(shim, "prepare_cfg")
    z_two = run(shim).double().cpu().numpy()
    z_bat = run(ld).double().cpu().numpy()
    e_ref = _rms(z_two - g["latent"]) / _rms(g["latent"])
    e_bat = _rms(z_two - z_bat) / _rms(z_bat)
    print(f"two-pass CFG fallback: latent rel rms vs reference {e_ref:.2e}, vs batched path {e_bat:.2e}")
    assert e_ref < 1e-4   # same bar as test_e2e_5step_matches_reference_generate_batch
    assert e_bat < 1e-5

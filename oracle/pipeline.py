"""Oracle: the whole text-to-audio sampling path on the CPU, assembled from the functional
restatements (oracle/unet.py, vae.py, ddim.py).  TEST INFRA ONLY — the checker for parity tests,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of bench.py; never on the product path.

Restates LatentDiffusion.generate_batch (models/ddpm.py:1477-1570) for n_gen = 1:
  get_input (ddpm.py:830-897): posterior sample of the encoded zero mel -> only its RNG draw matters,
  conditioners -> cond dict, unconditional conds, sample_log -> DDIMSampler.sample,
  decode_first_stage (ddpm.py:922-926), mel_spectrogram_to_waveform (ddpm.py:928-939),
  apply_model / DiffusionWrapper.forward routing (ddpm.py:1034-1042, 1821-1879).
"""
from __future__ import annotations

import json
import os
from typing import Dict, Optional

import torch

from . import cases, weights
from .ddim import ancestral_sample, ddim_sample, make_schedule_buffers
from .unet import unet_forward
from .vae import hifigan_forward, vae_decode, vae_encode_moments

_GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def hot_path_shapes(model_name: str = "audioldm2-full") -> Dict[str, tuple]:
    """Names/shapes of the reference's hot-path tensors (recorded from the real LatentDiffusion)."""
    fname = {"audioldm2-full": "e2e_statedict_keys.json", "audioldm_48k": "e2e48k_statedict_keys.json",
             "audioldm2-speech-gigaspeech": "e2espeech_statedict_keys.json",
             "audioldm2-full-large-1150k": "e2elarge_statedict_keys.json"}[model_name]
    with open(os.path.join(_GOLD, fname)) as f:
        return {k: tuple(v) for k, v in json.load(f).items()}


def oracle_48k(seed: int = 0) -> "OracleLatentDiffusion":
    """BASELINE config 3 (audioldm_48k): FiLM-conditioned UNet on a [16, 128, 32] latent, 4-level VAE,
    48 kHz HiFi-GAN (utils.py:413-561)."""
    from audioldm2_amd.pipeline import default_audioldm_config
    o = OracleLatentDiffusion(sd=weights.make_state_dict(hot_path_shapes("audioldm_48k"), seed=seed),
                              unet_cfg=cases.UNET_48K, ddconfig=cases.DDCONFIG_48K, hifigan_cfg=cases.HIFIGAN_48K,
                              cond_cfg=default_audioldm_config("audioldm_48k")["model"]["params"]["cond_stage_config"])
    o.channels, o.latent_t_size, o.latent_f_size = 16, 128, 32
    return o


def oracle_named(model_name: str, seed: int = 0) -> "OracleLatentDiffusion":
    """BASELINE configs 4 / 5: `audioldm2-full-large-1150k` (context_dim [768, 1024, None], transformer depth 2,
    utils.py:118-120) and `audioldm2-speech-gigaspeech` (one 512-token context, utils.py:121-187); 16 kHz VAE and
    vocoder as in the full model.  The UNet config is the reference's (audioldm2_amd.pipeline.default_audioldm_config
    reproduces utils.py's dicts and is checked against them in tests/test_host_logic.py)."""
    from audioldm2_amd.pipeline import default_audioldm_config
    params = default_audioldm_config(model_name)["model"]["params"]
    return OracleLatentDiffusion(sd=weights.make_state_dict(hot_path_shapes(model_name), seed=seed),
                                 unet_cfg=params["unet_config"]["params"], cond_cfg=params["cond_stage_config"])


class OracleLatentDiffusion:
    def __init__(self, sd: Optional[Dict[str, torch.Tensor]] = None, unet_cfg=None, ddconfig=None,
                 hifigan_cfg=None, scale_factor: float = cases.SCALE_FACTOR, t5_len: int = 32, seed: int = 0,
                 cond_cfg=None):
        self.unet_cfg = unet_cfg or cases.UNET_FULL
        self.dd = ddconfig or cases.DDCONFIG_16K
        self.hcfg = hifigan_cfg or cases.HIFIGAN_16K
        self.sd = sd if sd is not None else weights.make_state_dict(hot_path_shapes(), seed=seed)
        self.scale_factor = scale_factor
        self.buffers = make_schedule_buffers(1000, 0.0015, 0.0195)
        from audioldm2_amd.pipeline import instantiate_from_config
        if cond_cfg is None:
            cond_cfg = cases.e2e_cond_config("cpu", t5_len)
        for v in cond_cfg.values():
            v["params"]["device"] = "cpu"
        self.cond_keys = list(cond_cfg.keys())
        self.cond_models = {k: instantiate_from_config(v) for k, v in cond_cfg.items()}
        self.cond_stage_key = {k: v["cond_stage_key"] for k, v in cond_cfg.items()}
        self.channels, self.latent_t_size, self.latent_f_size = 8, 256, 16
        self.conditional_dry_run_finished = False   # ddpm.py:679

    def _cfg_dropout_draw(self):
        """ddpm.py:850-855, 916-917: `get_input` draws `torch.rand(1)` (make_decision(unconditional_prob_cfg), always "no" at p = 0.0)
        on every call except the first of the object's life — between the posterior sample and the conditioners."""
        if self.conditional_dry_run_finished and len(self.cond_keys) > 0:   # ddpm.py:850: `if len(cond_stage_model_metadata) > 0`
            torch.rand(1)
        self.conditional_dry_run_finished = True   # ddpm.py:916: unconditionally

    def apply_model(self, x, t, cond):
        # DiffusionWrapper.forward (ddpm.py:1821-1879): film* keys -> y (squeeze(1), concatenated),
        # crossattn* keys -> (context, mask) in config-key order
        y, ctxs, masks = None, [], []
        for k in self.cond_keys:
            if "film" in k:
                v = cond[k].squeeze(1)
                y = v if y is None else torch.cat([y, v], dim=-1)
            else:
                ctxs.append(cond[k][0])
                masks.append(cond[k][1])
        return unet_forward(self.sd, self.unet_cfg, x, t, ctxs, masks, y=y, prefix="model.diffusion_model.")

    @torch.no_grad()
    def generate_batch(self, batch, unconditional_guidance_scale=3.5, ddim_steps=200, ddim_eta=1.0,
                       record: Optional[list] = None):
        B = batch["log_mel_spec"].shape[0]
        f = 2 ** (len(self.dd["ch_mult"]) - 1)
        torch.randn((B, self.dd["z_channels"], batch["log_mel_spec"].shape[1] // f,
                     batch["log_mel_spec"].shape[2] // f))  # posterior draw (distributions.py:37-41)
        self._cfg_dropout_draw()
        cond = {k: m(batch if self.cond_stage_key[k] == "all" else batch[self.cond_stage_key[k]])
                for k, m in self.cond_models.items()}
        uncond = None
        if unconditional_guidance_scale != 1.0:
            uncond = {k: m.get_unconditional_condition(B) for k, m in self.cond_models.items()}
        shape = (B, self.channels, self.latent_t_size, self.latent_f_size)
        z = ddim_sample(self.apply_model, shape, cond, uncond, unconditional_guidance_scale, ddim_steps,
                        ddim_eta, self.buffers["alphas_cumprod"], record=record)
        zs = (1.0 / torch.tensor(self.scale_factor)) * z  # ddpm.py:924 (fp32 buffer arithmetic)
        mel = vae_decode(self.sd, self.dd, zs, prefix="first_stage_model.")
        wave = hifigan_forward(self.sd, self.hcfg, mel.squeeze(1).permute(0, 2, 1),
                               prefix="first_stage_model.vocoder.")
        return {"latent": z, "mel": mel, "wave": wave.numpy()}

    @torch.no_grad()
    def generate_batch_masked(self, batch, unconditional_guidance_scale=2.5, ddim_steps=200, ddim_eta=1.0,
                              time_mask_ratio_start_and_end=(0.25, 0.75), freq_mask_ratio_start_and_end=(0.75, 1.0)):
        """ddpm.py:1573-1676 for n_gen = 1: encode the mel (posterior sample on the host generator,
        distributions.py:37-41; x scale_factor, ddpm.py:793-802), mask, masked DDIM, decode, vocode."""
        x = batch["log_mel_spec"].unsqueeze(1).float()
        moments = vae_encode_moments(self.sd, self.dd, x, prefix="first_stage_model.")
        mean, logvar = torch.chunk(moments, 2, dim=1)
        std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))
        z = torch.tensor(self.scale_factor) * (mean + std * torch.randn(mean.shape))
        B, _, h, w = z.shape
        self._cfg_dropout_draw()
        cond = {k: m(batch if self.cond_stage_key[k] == "all" else batch[self.cond_stage_key[k]])
                for k, m in self.cond_models.items()}
        mask = torch.ones(B, h, w)
        mask[:, int(h * time_mask_ratio_start_and_end[0]):int(h * time_mask_ratio_start_and_end[1]), :] = 0
        mask[:, :, int(w * freq_mask_ratio_start_and_end[0]):int(w * freq_mask_ratio_start_and_end[1])] = 0
        mask = mask[:, None]
        uncond = None
        if unconditional_guidance_scale != 1.0:
            uncond = {k: m.get_unconditional_condition(B) for k, m in self.cond_models.items()}
        lat = ddim_sample(self.apply_model, (B, self.channels, h, w), cond, uncond, unconditional_guidance_scale,
                          ddim_steps, ddim_eta, self.buffers["alphas_cumprod"], mask=mask, x0=z)
        mel = vae_decode(self.sd, self.dd, (1.0 / torch.tensor(self.scale_factor)) * lat, prefix="first_stage_model.")
        wave = hifigan_forward(self.sd, self.hcfg, mel.squeeze(1).permute(0, 2, 1),
                               prefix="first_stage_model.vocoder.")
        return {"x0": z, "mask": mask, "latent": lat, "mel": mel, "wave": wave.numpy()}

    @torch.no_grad()
    def sample_ancestral(self, batch, timesteps: int):
        """LatentDiffusion.sample (ddpm.py:1350-1391) for the last `timesteps` steps, no CFG."""
        B = len(batch["text"])
        cond = {k: m(batch if self.cond_stage_key[k] == "all" else batch[self.cond_stage_key[k]])
                for k, m in self.cond_models.items()}
        shape = (B, self.channels, self.latent_t_size, self.latent_f_size)
        return ancestral_sample(self.apply_model, shape, cond, timesteps)

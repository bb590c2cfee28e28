"""Import the REAL reference modules from /root/reference (build container only).  TEST INFRA ONLY.

Recipe of SURVEY.md Appendix A: the reference's hot path is pure PyTorch, but its package
`__init__` and a few module headers import packages that are not installed here (soundfile,
progressbar, librosa, torchaudio, torchvision, ...).  None of them is *called* on the hot path, so
import-time MagicMock stubs are enough.  Nothing is copied: the reference code runs where it lies.
"""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import os
import sys
import types
from unittest.mock import MagicMock

REF_ROOT = os.environ.get("ALDM_REFERENCE_ROOT", "/root/reference")
_STUB_ROOTS = {"soundfile", "progressbar", "librosa", "torchaudio", "torchvision", "timm", "torchlibrosa",
               "phonemizer", "unidecode", "ftfy", "chardet", "gradio", "ipdb", "pytorch_lightning",
               "taming", "kornia", "wandb", "matplotlib", "soxr"}


class _StubLoader(importlib.abc.Loader):
    def create_module(self, spec):
        m = MagicMock(name=spec.name)
        m.__path__ = []
        m.__name__ = spec.name
        m.__spec__ = spec
        m.__loader__ = self
        return m

    def exec_module(self, module):
        return None


class _StubFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path, target=None):
        root = fullname.split(".")[0]
        if root in _STUB_ROOTS:
            try:  # prefer the real package when it IS installed
                for f in sys.meta_path:
                    if f is self:
                        continue
                    spec = f.find_spec(fullname, path, target) if hasattr(f, "find_spec") else None
                    if spec is not None:
                        return spec
            except Exception:
                pass
            return importlib.machinery.ModuleSpec(fullname, _StubLoader(), is_package=True)
        return None


_installed = False


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "audioldm2"))


def install():
    """Make `import audioldm2.<hot-path module>` work against /root/reference."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError(f"reference not found under {REF_ROOT}")
    sys.meta_path.insert(0, _StubFinder())
    pkg = types.ModuleType("audioldm2")
    pkg.__path__ = [os.path.join(REF_ROOT, "audioldm2")]
    sys.modules["audioldm2"] = pkg
    # ddpm.py:12 does `from ...encoders.modules import *` and names CLAPAudioEmbeddingClassifierFreev2
    # at ddpm.py:114; the real module needs Hub tokenizers, so a tiny stand-in is pre-seeded.
    import torch.nn as nn

    enc = types.ModuleType("audioldm2.latent_diffusion.modules.encoders.modules")

    class CLAPAudioEmbeddingClassifierFreev2(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    enc.CLAPAudioEmbeddingClassifierFreev2 = CLAPAudioEmbeddingClassifierFreev2
    enc.__all__ = ["CLAPAudioEmbeddingClassifierFreev2"]
    sys.modules["audioldm2.latent_diffusion.modules.encoders.modules"] = enc
    _installed = True


def unet_cls():
    install()
    from audioldm2.latent_diffusion.modules.diffusionmodules.openaimodel import UNetModel
    return UNetModel


def vae_decoder_encoder():
    install()
    from audioldm2.latent_diffusion.modules.diffusionmodules.model import Decoder, Encoder
    return Decoder, Encoder


def hifigan_generator(cfg: dict):
    install()
    import audioldm2.hifigan as hifigan
    g = hifigan.Generator_old(hifigan.AttrDict(cfg))
    g.eval()
    g.remove_weight_norm()
    return g


def ddim_sampler_cls():
    install()
    from audioldm2.latent_diffusion.models.ddim import DDIMSampler
    return DDIMSampler


def sequence_generator(sequence_gen_length: int, sequence_input_key, sequence_input_embed_dim):
    """The real `Sequence2AudioMAE` (audioldm2/audiomae_gen/sequence_input.py) without conditioner sub-modules (the hot
    function `generate` takes an explicit cond_dict).  `GPT2Config.from_pretrained("gpt2")` (sequence_input.py:69) needs
    the Hub; GPT2Config()'s defaults ARE that configuration (768 / 12 layers / 12 heads / 1024 positions / gelu_new),
    so it is patched to return them."""
    install()
    from transformers import GPT2Config
    GPT2Config.from_pretrained = classmethod(lambda cls, name, *a, **k: GPT2Config())
    from audioldm2.audiomae_gen.sequence_input import Sequence2AudioMAE
    return Sequence2AudioMAE(base_learning_rate=2e-4, sequence_gen_length=sequence_gen_length,
                             sequence_input_key=list(sequence_input_key),
                             sequence_input_embed_dim=list(sequence_input_embed_dim), cond_stage_config={},
                             batchsize=16).eval()


def phoneme_encoder(vocabs_size: int, pad_length: int, pad_token_id: int):
    """The real `PhonemeEncoder` (audioldm2/latent_diffusion/modules/encoders/modules.py:30-110).  refimport pre-seeds a
    stand-in for that module (it pulls Hub tokenizers at import), so the class is taken from the file itself, loaded
    under a private name with the un-installable imports stubbed."""
    install()
    import importlib.util
    import sys
    path = os.path.join(REF_ROOT, "audioldm2", "latent_diffusion", "modules", "encoders", "modules.py")
    name = "_aldm_ref_encoders_modules"
    if name not in sys.modules:
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
    return sys.modules[name].PhonemeEncoder(vocabs_size=vocabs_size, pad_length=pad_length, pad_token_id=pad_token_id).eval()


def flan_t5_hidden_state(cfg: dict, token_batch):
    """The real `FlanT5HiddenState` (encoders/modules.py:113-198) with its two Hub calls replaced: `T5Config.from_pretrained`
    returns T5Config(**cfg) and `AutoTokenizer.from_pretrained` a stand-in whose __call__ returns `token_batch`
    (input_ids, attention_mask) — the sentencepiece model is not reachable offline, so the oracle's boundary is token ids."""
    install()
    import importlib.util
    import sys
    import types
    from transformers import T5Config
    path = os.path.join(REF_ROOT, "audioldm2", "latent_diffusion", "modules", "encoders", "modules.py")
    name = "_aldm_ref_encoders_modules"
    if name not in sys.modules:
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
    mod = sys.modules[name]

    class _Tok:
        def __call__(self, prompt, **kw):
            ids, mask = token_batch(prompt)
            return types.SimpleNamespace(input_ids=ids, attention_mask=mask)

    class _Cfg:
        @staticmethod
        def from_pretrained(name, *a, **k):
            return T5Config(**cfg)

    class _AutoTok:
        @staticmethod
        def from_pretrained(name, *a, **k):
            return _Tok()
    saved = mod.T5Config, mod.AutoTokenizer
    mod.T5Config, mod.AutoTokenizer = _Cfg, _AutoTok
    try:
        m = mod.FlanT5HiddenState()
    finally:
        mod.T5Config, mod.AutoTokenizer = saved
    return m.eval()


def htsat_swin_transformer(hcfg: dict, acfg: dict):
    """The REAL `HTSAT_Swin_Transformer` (audioldm2/clap/open_clip/htsat.py:777) at geometry `hcfg`, in eval mode, with its two
    torchlibrosa extractors (a package that is neither vendored nor installed) replaced by the restated ones of
    oracle/htsat.py; everything downstream of the log-mel — bn0, reshape_wav2img, patch embedding, every Swin block, patch
    merging, pooling — is the reference's own code."""
    install()
    import importlib
    import types
    import torch.nn as nn
    from . import htsat as oh
    ht = importlib.import_module("audioldm2.clap.open_clip.htsat")
    cfg = types.SimpleNamespace(audio_length=1024, clip_samples=acfg["clip_samples"], mel_bins=acfg["mel_bins"],
                                sample_rate=acfg["sample_rate"], window_size=acfg["window_size"], hop_size=acfg["hop_size"],
                                fmin=acfg["fmin"], fmax=acfg["fmax"], class_num=527, model_type="HTSAT", model_name="base")
    m = ht.HTSAT_Swin_Transformer(spec_size=hcfg["spec_size"], patch_size=hcfg["patch"], patch_stride=(hcfg["patch"],) * 2,
                                  num_classes=527, embed_dim=hcfg["embed_dim"], depths=list(hcfg["depths"]),
                                  num_heads=list(hcfg["num_heads"]), window_size=hcfg["window_size"], config=cfg,
                                  enable_fusion=False, fusion_type="None").eval()

    class Spec(nn.Module):   # torchlibrosa.stft.Spectrogram: (B, T) -> (B, 1, frames, freq)
        def forward(self, x):
            return oh.power_spectrogram(x, acfg["window_size"], acfg["hop_size"])[:, None]

    class LogMel(nn.Module):  # torchlibrosa.stft.LogmelFilterBank: (B, 1, frames, freq) -> (B, 1, frames, mel)
        def forward(self, x):
            return oh.logmel(x[:, 0], acfg)[:, None]
    m.spectrogram_extractor, m.logmel_extractor = Spec(), LogMel()
    return m

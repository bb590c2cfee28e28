"""VITS phoneme encoder on the MI355X (second half; BASELINE config 5 `crossattn_vits_phoneme`).

Drop-in for `audioldm2.latent_diffusion.modules.encoders.modules.PhonemeEncoder` (encoders/modules.py:30-110): same
constructor keywords, same state-dict keys (`learnable_positional_embedding`, `text_encoder.emb.weight`,
`text_encoder.encoder.{attn_layers.i.{conv_q,conv_k,conv_v,conv_o,emb_rel_k,emb_rel_v}, norm_layers_{1,2}.i.{gamma,beta},
ffn_layers.i.{conv_1,conv_2}}`, `text_encoder.proj`), same `forward(phoneme_idx) -> [emb [B, T, 192], mask [B, T]]` and
`get_unconditional_condition(batchsize)` — usable through `cond_stage_config[...].target` (utils.py:156-165).

The nn children only hold parameters; forward() runs channels-last tokens [B, T, 192] on the kernel library: q/k/v as ONE
fused GEMM (aldm_igemm), the windowed relative-position attention as one kernel (aldm_rel_attention: the reference's
pad / reshape skews are index shifts there), the k = 3 conv FFN as implicit GEMMs with the ReLU in the epilogue, LayerNorm
over channels = aldm_layernorm on the channels-last rows.  torch only gathers the embedding rows.  No CPU fallback.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import ops

# device of the packed weights and activations; tests/test_host_logic.py points it at the CPU next to torch stand-ins for `ops`
_DEV = torch.device("cuda")


class _LayerNorm(nn.Module):
    """attentions.py:11-23 (parameters named gamma / beta)."""

    def __init__(self, channels):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(channels))
        self.beta = nn.Parameter(torch.zeros(channels))


class _MultiHeadAttention(nn.Module):
    """attentions.py:183-236 parameter holder (window_size set, heads_share)."""

    def __init__(self, channels, n_heads, window_size):
        super().__init__()
        self.n_heads = n_heads
        k_channels = channels // n_heads
        self.conv_q = nn.Conv1d(channels, channels, 1)
        self.conv_k = nn.Conv1d(channels, channels, 1)
        self.conv_v = nn.Conv1d(channels, channels, 1)
        self.conv_o = nn.Conv1d(channels, channels, 1)
        self.emb_rel_k = nn.Parameter(torch.randn(1, window_size * 2 + 1, k_channels) * k_channels ** -0.5)
        self.emb_rel_v = nn.Parameter(torch.randn(1, window_size * 2 + 1, k_channels) * k_channels ** -0.5)


class _FFN(nn.Module):
    """attentions.py:374-430 parameter holder (non-causal, relu)."""

    def __init__(self, channels, filter_channels, kernel_size):
        super().__init__()
        self.conv_1 = nn.Conv1d(channels, filter_channels, kernel_size)
        self.conv_2 = nn.Conv1d(filter_channels, channels, kernel_size)


class _Encoder(nn.Module):
    """attentions.py:26-87"""

    def __init__(self, hidden_channels, filter_channels, n_heads, n_layers, kernel_size, window_size=4):
        super().__init__()
        self.attn_layers = nn.ModuleList(_MultiHeadAttention(hidden_channels, n_heads, window_size) for _ in range(n_layers))
        self.norm_layers_1 = nn.ModuleList(_LayerNorm(hidden_channels) for _ in range(n_layers))
        self.ffn_layers = nn.ModuleList(_FFN(hidden_channels, filter_channels, kernel_size) for _ in range(n_layers))
        self.norm_layers_2 = nn.ModuleList(_LayerNorm(hidden_channels) for _ in range(n_layers))


class TextEncoder(nn.Module):
    """phoneme_encoder/encoder.py:9-37 parameter holder."""

    def __init__(self, n_vocab, out_channels=192, hidden_channels=192, filter_channels=768, n_heads=2, n_layers=6,
                 kernel_size=3, p_dropout=0.1):
        super().__init__()
        self.hidden_channels, self.n_heads, self.kernel_size = hidden_channels, n_heads, kernel_size
        self.emb = nn.Embedding(n_vocab, hidden_channels)
        nn.init.normal_(self.emb.weight, 0.0, hidden_channels ** -0.5)
        self.encoder = _Encoder(hidden_channels, filter_channels, n_heads, n_layers, kernel_size)
        self.proj = nn.Conv1d(hidden_channels, out_channels * 2, 1)  # (m, logs): not used by PhonemeEncoder.forward


class PhonemeEncoder(nn.Module):
    def __init__(self, vocabs_size=41, pad_length=250, pad_token_id=None):
        super().__init__()
        assert pad_token_id is not None
        self.device = None
        self.PAD_LENGTH = int(pad_length)
        self.pad_token_id = pad_token_id
        self.text_encoder = TextEncoder(n_vocab=vocabs_size, out_channels=192, hidden_channels=192, filter_channels=768,
                                        n_heads=2, n_layers=6, kernel_size=3, p_dropout=0.1)
        self.learnable_positional_embedding = nn.Parameter(torch.zeros((1, 192, self.PAD_LENGTH)))
        self._pk = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_packed())

    def invalidate_packed(self):
        self._pk = None

    def _prepare(self):
        if self._pk is None:
            f = lambda t: t.detach().float().to(_DEV).contiguous()
            te = self.text_encoder
            enc = te.encoder
            layers = []
            for a, n1, ff, n2 in zip(enc.attn_layers, enc.norm_layers_1, enc.ffn_layers, enc.norm_layers_2):
                layers.append(dict(
                    qkv=ops.pack_conv(torch.cat([a.conv_q.weight, a.conv_k.weight, a.conv_v.weight], 0)[:, :, 0],
                                      torch.cat([a.conv_q.bias, a.conv_k.bias, a.conv_v.bias], 0)),
                    o=ops.pack_conv(a.conv_o.weight[:, :, 0], a.conv_o.bias),
                    ek=f(a.emb_rel_k[0]), ev=f(a.emb_rel_v[0]),
                    ln1=(f(n1.gamma), f(n1.beta)), ln2=(f(n2.gamma), f(n2.beta)),
                    c1=ops.pack_conv(ff.conv_1.weight, ff.conv_1.bias), c2=ops.pack_conv(ff.conv_2.weight, ff.conv_2.bias)))
            self._pk = dict(
                # emb(x) * sqrt(hidden) (encoder.py:40): scaling the table once is the same fp32 product per element
                table=f(te.emb.weight) * math.sqrt(te.hidden_channels),
                pos=f(self.learnable_positional_embedding[0].t()),   # [T, 192] channels-last
                layers=layers)
        return self._pk

    def get_unconditional_condition(self, batchsize):
        """encoders/modules.py:63-67: the encoder on an all-pad sequence."""
        return self(torch.full((batchsize, self.PAD_LENGTH), self.pad_token_id, dtype=torch.long))

    @torch.no_grad()
    def forward(self, phoneme_idx):
        """encoders/modules.py:94-110.  phoneme_idx: [B, T] token ids (pads at the end).  Returns
        [text_emb [B, T, 192], mask [B, T]] (fp32, on the GPU)."""
        pk = self._prepare()
        dev = pk["table"].device
        idx = phoneme_idx.to(dev)
        B, T = idx.shape
        assert T == pk["pos"].shape[0], f"sequence length {T} != pad_length {pk['pos'].shape[0]}"
        length = (idx != self.pad_token_id).sum(-1)                                        # modules.py:75-82
        mask = (torch.arange(T, device=dev)[None, :] < length[:, None]).float().contiguous()   # commons.sequence_mask
        H = self.text_encoder.n_heads
        pad = self.text_encoder.kernel_size // 2
        x = pk["table"].index_select(0, idx.reshape(-1)).view(B, T, -1)                    # encoder.py:40
        x = ops.rowscale_add(x, mask)                                                      # encoder.py:46, attentions.py:77
        C = x.shape[-1]
        for L in pk["layers"]:
            qkv = ops.linear(x, L["qkv"])
            y = ops.rel_attention(qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:], H, L["ek"], L["ev"], mask)
            x = ops.layernorm(ops.linear(y, L["o"], res=x), *L["ln1"])                     # attentions.py:79-81
            h = ops.conv(ops.rowscale_add(x, mask).view(B, 1, T, C), L["c1"], pad=(0, pad), act=ops.ACT_LRELU,
                         act_slope=0.0)                                                    # :406-410 (relu)
            h = ops.conv(ops.rowscale_add(h, mask), L["c2"], pad=(0, pad)).view(B, T, C)   # :412
            x = ops.layernorm(ops.rowscale_add(h, mask, res=x), *L["ln2"])                 # :413, :83-85
        out = ops.rowscale_add(x, mask, res=pk["pos"].unsqueeze(0).expand(B, T, C).contiguous())   # :86, modules.py:103
        return [out, mask]


# ---- phoneme string -> ids (latent_diffusion/util.py:14-49) --------------------------------------------------------------
# The VITS symbol inventory the speech checkpoints were trained on: pad, punctuation, Latin letters, IPA letters, specials —
# 183 entries = PhonemeEncoder(vocabs_size=183).  The apostrophe occurs twice in the IPA block; like the reference's dict
# comprehension the LAST occurrence wins.
VITS_PAD_LENGTH = 310
VITS_SYMBOLS = (["_"] + list(';:,.!?¡¿—…"«»“” ') + list("ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz") +
                list("ɑɐɒæɓʙβɔɕçɗɖðʤəɘɚɛɜɝɞɟʄɡɠɢʛɦɧħɥʜɨɪʝɭɬɫɮʟɱɯɰŋɳɲɴøɵɸθœɶʘɹɺɾɻʀʁɽʂʃʈʧʉʊʋⱱʌɣɤʍχʎʏʑʐʒʔʡʕʢǀǁǂǃˈˌːˑʼʴʰʱʲʷˠˤ˞↓↑→↗↘'̩'ᵻ") +
                list("♪☎☒☝⚠"))
_VITS_SYMBOL_TO_ID = {sym: i for i, sym in enumerate(VITS_SYMBOLS)}


def phoneme_ids(phonemes, batchsize: int = 1, pad_length: int = VITS_PAD_LENGTH) -> torch.Tensor:
    """`phoneme_idx` [batchsize, pad_length] of an ALREADY phonemised transcription (an IPA string, what the reference's
    `text2phoneme` = phonemizer / espeak front end returns; that front end is out of scope here): the string plus the end
    mark, one id per character, unknown characters -> the pad symbol "_" (with the reference's message), cut / zero-padded
    to pad_length, the same row for every sample — latent_diffusion/util.py:28-49."""
    seq = []
    text = phonemes + "⚠"
    for sym in text:
        if sym not in _VITS_SYMBOL_TO_ID:
            print("%s is not in the vocabulary. %s" % (sym, text))
            sym = "_"
        seq.append(_VITS_SYMBOL_TO_ID[sym])
    seq = seq[:pad_length]
    seq = seq + [0] * (pad_length - len(seq))
    return torch.LongTensor(seq).unsqueeze(0).expand(batchsize, -1)

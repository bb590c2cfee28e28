"""oracle/ — CPU restatement of the reference's sampling-path arithmetic.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this package,
and only as the checker / the timed CPU baseline — never from the product (`audioldm2_amd/`), which
has no CPU fallback and fails loudly when libaldm_hip.so is missing.

What it is: plain PyTorch fp32 (CPU) functional code, written from the reference's behaviour, each
function citing the `/root/reference/audioldm2/...` file:line it restates.  The reference path is
floating-point, so a torch fp32 restatement is the appropriate oracle (numpy for the STFT bases and
the mel filterbank).

How it is pinned (the reference ships no tests, no golden vectors — SURVEY.md §4/§8c):
  * `oracle/make_golden.py` imports the REAL reference modules from /root/reference in the build
    container (import stubs of SURVEY.md Appendix A), loads the same deterministic weights
    (`oracle/weights.py`) and writes input/output fixtures to `tests/golden/*.npz`;
  * `tests/test_oracle.py` checks this restatement against those fixtures on the CPU
    (runs everywhere, no reference needed) and `tests/test_oracle_vs_reference.py` checks it
    against the live reference classes when /root/reference is present.
  * third-party arithmetic not vendored by the reference: `librosa==0.9.2` mel filterbank /
    pad_center (stft.py:5-6,42,145-147) is restated from its published definition in
    `oracle/stft.py` — parity unpinned for that piece (no librosa offline, no reference test); the restated basis is
    cross-checked against transformers.audio_utils.mel_filter_bank, an independent restatement of the same definition.
  * `oracle/seqgen.py` (the AudioMAE-token sequence generator): restates sequence_input.py and the
    un-vendored `transformers==4.30.2` GPT2Model it drives; pinned by fixtures from the REAL reference class running on
    the installed transformers (5.x — same GPT-2 arithmetic), `GPT2Config.from_pretrained("gpt2")` replaced by the
    identical default GPT2Config() because the Hub is unreachable.
  * `oracle/bf16x6.py` is not a restatement of the reference but of how the HIP igemm kernels evaluate fp32 products
    on the bf16 matrix cores (exact operand split, six partial products) and of the split weight image's layout.
"""

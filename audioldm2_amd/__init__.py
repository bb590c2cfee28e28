"""audioldm2_amd — MI355X-native (gfx950) AudioLDM2 sampling hot path.

DDIM loop / UNet -> VAE decoder -> HiFi-GAN vocoder (+ STFT/mel front-end) as hand-written HIP
kernels behind a C ABI (include/aldm_hip.h, libaldm_hip.so); the Python modules here mirror the
reference's plugin interface (same constructor kwargs, same state-dict keys) so they drop in via
the reference's `target:` config strings.  See DESIGN.md and INTEGRATION.md.
"""
__version__ = "0.1.0"

"""A few launches of the 1024x1024 self-attention (batch 16, 8 heads, current matrix-core mode) for a rocprofv3 --pmc pass."""
import sys

import torch

sys.path.insert(0, ".")
from audioldm2_amd import ops  # noqa: E402

g = torch.Generator().manual_seed(0)
B, heads, L = 16, 8, 1024
C = heads * 32
qkv = torch.randn(B, L, 3 * C, generator=g).cuda()
for _ in range(5):
    o = ops.attention(qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:], heads)
torch.cuda.synchronize()
print("ok", float(o.abs().mean()))

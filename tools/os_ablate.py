"""Phase ablation of the operand-stationary DMA GEMM: the library named by $ALDM_LIB_PATH (an ALDM_OS_ABLATE build of
tools/gpu/build_variant.sh: wrong results by construction) on three shapes, HIP-graph timed."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm2_amd import ops  # noqa: E402
from tools.os_probe import Case, graph_time  # noqa: E402

tag = os.path.basename(os.environ.get("ALDM_LIB_PATH", "libaldm_hip.so")).replace("libaldm_", "").replace(".so", "")
R = 16
cases = [Case("geglu 16384x256->2x1024", "geglu", R * 1024, 256, 2048, split_out="only"),
         Case("qkv 16384x256->768", "qkv", R * 1024, 256, 768, bias=False, heads=8, L=1024),
         Case("proj 16384x256->256+res", "linear", R * 1024, 256, 256, res=True)]
st = 3 if ops.split_parts() == 3 else 4
print(f"{tag:12s}", "  ".join(f"{c.name}: {graph_time(lambda: c.run((32, 128, 300 + st))):6.1f}" for c in cases), flush=True)

"""Phase timeline of ONE block of the igemm kernel (instrumented debug build, s_memtime stamps written
through the workspace pointer): where do the cycles of a k-tile go?  Debug tool, GPU box only.
Usage: ALDM_LIB_PATH=tools/gpu/libaldm_trace.so python tools/igemm_trace.py"""
import ctypes as C
import os
import sys

import torch

os.environ["ALDM_NO_TUNING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from audioldm2_amd import lib as L  # noqa: E402
from audioldm2_amd import ops  # noqa: E402
import igemm_autotune as AT  # noqa: E402


def run(key, bm, bn, kg, cold):
    lib = L.load()
    d, keep, M, N, K = AT.make_desc(key)
    tr = torch.zeros(1024, dtype=torch.int64, device="cuda")
    d.ws = tr.data_ptr()
    d.ws_floats = 16
    d.hint_bm, d.hint_bn, d.hint_splits, d.hint_kgroups = bm, bn, 1, kg
    st = torch.cuda.current_stream().cuda_stream
    junk = torch.empty(512 << 20, dtype=torch.float32, device="cuda") if cold else None
    for _ in range(3):
        if cold:
            junk.fill_(1.0)
        tr.zero_()
        torch.cuda.synchronize()
        assert lib.aldm_igemm(C.byref(d), st) == 0, lib.aldm_last_error()
        torch.cuda.synchronize()
    t = tr.cpu().tolist()
    t = [x for x in t if x]
    nk = (K + 31) // 32
    n_it = (nk + kg - 1) // kg
    print(f"--- key M={M} N={N} K={K} tile {bm}x{bn} kg={kg} cold={cold}: {len(t)} stamps, total {t[-1]-t[0]} ticks")
    base = t[0]
    pro = [x - base for x in t[:4]]
    print(f"  prologue: start->loads issued {pro[1]}, ->committed {pro[2]}, ->barrier done {pro[3]}")
    its = t[4:4 + 5 * n_it]
    names = ["issue", "mma0", "commit", "mma1", "barrier"]
    acc = [0] * 5
    prev = t[3]
    rows = []
    for i in range(n_it):
        seg = its[5 * i:5 * i + 5]
        if len(seg) < 5:
            break
        dl = [seg[0] - prev] + [seg[j] - seg[j - 1] for j in range(1, 5)]
        prev = seg[4]
        rows.append(dl)
    for r in rows[:4]:
        print("  it:", " ".join(f"{n}={v}" for n, v in zip(names, r)), " sum", sum(r))
    if len(rows) > 2:
        mid = rows[1:-1] or rows
        avg = [sum(r[j] for r in mid) / len(mid) for j in range(5)]
        print("  avg(mid):", " ".join(f"{n}={v:.0f}" for n, v in zip(names, avg)), " sum", f"{sum(avg):.0f}")
    rest = t[4 + 5 * n_it:]
    if rest:
        print(f"  after loop: {[x - prev for x in rest]}")


def main():
    print(torch.cuda.get_device_name(0))
    keys = {
        "lin 1024x640x640": "1,1,1024,640,0,0,1,1,1,1,1,1,0,0,1,1,1,1024,640,0,1,0,0,0",
        "lin 4096x384x384": "1,1,4096,384,0,0,1,1,1,1,1,1,0,0,1,1,1,4096,384,0,1,0,0,0",
        "lin 16384x2048(geglu-free)x256": "1,1,16384,256,0,0,1,1,1,1,1,1,0,0,1,1,1,16384,2048,0,1,0,0,0",
        "conv 256x16 128->128 pre": "16,256,16,128,0,0,1,1,3,3,1,1,1,1,1,1,256,16,128,0,1,0,0,2",
    }
    for name, key in keys.items():
        print("====", name)
        for (bm, bn, kg) in ([(64, 64, 1), (64, 64, 2)] if "lin 1024" in name or "lin 4096" in name else [(128, 128, 1)]):
            for cold in (False, True):
                run(key, bm, bn, kg, cold)


if __name__ == "__main__":
    main()

#!/bin/bash
# round 5, call 6: deeper LDS rings (6 k-tiles) of the 64x64 loader-wave tile on the 1024-row launches IN THE STEP, where the weights
# arrive HBM-cold (the per-launch probe of call 3 measured them warm: no gain there) — same-box step A/B through a second table directory
O=gpurun_out/r5_6; mkdir -p $O /tmp/tab206 /tmp/tab204; export TMPDIR=/tmp
python - <<'PY'
import json
for depth, out in ((206, "/tmp/tab206"), (204, "/tmp/tab204")):
    t = json.load(open("audioldm2_amd/tuning/mi355x_igemm_dma.json"))
    n = 0
    for k, v in t["entries"].items():
        f = k.split(",")
        rows = int(f[0]) * int(f[16]) * int(f[17])   # B * OH * OW
        if v[0] == 64 and v[1] == 64 and v[3] == 203 and rows <= 1024 and v[2] == 1:
            v[3] = depth
            n += 1
    json.dump(t, open(out + "/mi355x_igemm_dma.json", "w"), indent=0, sort_keys=True)
    print(depth, "entries changed:", n)
PY
{
for i in 1 2; do
timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/ring 3 (shipped): /'
ALDM_TUNING_DIR=/tmp/tab204 timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/ring 4: /'
ALDM_TUNING_DIR=/tmp/tab206 timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/ring 6: /'
done
} > $O/step_ab_ring.txt 2>&1; cat $O/step_ab_ring.txt

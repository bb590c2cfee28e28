#!/bin/bash
# round 6, call 15: rocprofv3 kernel statistics + gaps of the split-K decode step (96 graph-replayed tokens)
O=gpurun_out/r6_15; mkdir -p $O; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
PYTHONPATH=$R PROBE_B=8 PROBE_MODES=split timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_dec -o dec --output-format csv -- python $R/tools/decode_probe.py 96 > $R/$O/probe.txt 2>&1
cd $R
cp $(find /tmp/prof_dec -name "*kernel_stats.csv" | head -1) $O/decode_kernel_stats.csv
head -16 $O/decode_kernel_stats.csv | cut -c1-200
python - <<'PY' > $O/decode_gaps.txt 2>&1
import csv, glob
rows=[]
for f in glob.glob('/tmp/prof_dec/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
# the last 40 tokens' worth of dispatches (graph replays): tokens are delimited by decode_reduce_ln launches with the LN_f... use windows of 93
tail = rows[-93*40:]
busy = sum(e-s for s,e,_ in tail); span = tail[-1][1]-tail[0][0]
print(f"last {len(tail)} dispatches: span {span/1e3:.1f} us, kernel time {busy/1e3:.1f} us, gaps {(span-busy)/1e3:.1f} us; per token (93 dispatches): span {span/40/1e3:.1f} us busy {busy/40/1e3:.1f} us")
from collections import defaultdict
d=defaultdict(lambda:[0,0])
for s,e,n in tail:
    k=n.split('(')[0][:60]; d[k][0]+=1; d[k][1]+=e-s
for k,(n,t) in sorted(d.items(), key=lambda kv:-kv[1][1]): print(f"{k:60s} {n/40:6.1f}/token {t/n/1e3:7.2f} us avg {t/40/1e3:8.1f} us/token")
gaps=[tail[i+1][0]-tail[i][1] for i in range(len(tail)-1)]
gaps.sort(); print("gap median %.2f us, p90 %.2f us, max %.2f us" % (gaps[len(gaps)//2]/1e3, gaps[int(len(gaps)*0.9)]/1e3, gaps[-1]/1e3))
PY
cat $O/decode_gaps.txt

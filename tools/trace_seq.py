#!/usr/bin/env python3
"""Prints the launches of the LAST whole step in a rocprofv3 --kernel-trace directory (a step = pack_weights to pack_weights): duration, gap to the
previous launch's end, name.  usage: tools/trace_seq.py <rocprofv3 output dir>"""
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# take the last 53 launches before the end
names=[(r['Kernel_Name'],int(r['End_Timestamp'])-int(r['Start_Timestamp']),int(r['Start_Timestamp'])) for r in rows]
# find last occurrence of pack kernel 'stack_pack' or pack_weights to delimit a step
idx=[i for i,n in enumerate(names) if 'pack_weights' in n[0]]
s=idx[-2]; e=idx[-1]
prev=None
for n,d,t in names[s:e]:
    gap = (t-prev)/1000 if prev is not None else 0
    print(f"{d/1000:8.1f} us  gap {gap:6.1f}  {n[:110]}")
    prev=t+d
print('launches', e-s, 'total kernels', sum(d for _,d,_ in names[s:e])/1000, 'us; wall', (names[e][2]-names[s][2])/1000)

"""Developer tool (GPU box): idle time between kernels in a rocprofv3 --kernel-trace CSV directory.
    rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python3 bench.py ...; python3 tools/trace_gaps.py /tmp/tr
Prints the span, the summed gaps, the largest ones and the kernels they precede (steady state: < 1 ms per frame of C4)."""
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# take the last 160*... just compute over all: busy time vs span
start=int(rows[0]['Start_Timestamp']); end=max(int(r['End_Timestamp']) for r in rows)
busy=0; cur_end=start; gaps=[]
for r in rows:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    if s>cur_end: gaps.append((s-cur_end, r['Kernel_Name'][:40]))
    cur_end=max(cur_end,e)
tot_gap=sum(g for g,_ in gaps)
print('dispatches',len(rows),'span ms',(end-start)/1e6,'gap ms',tot_gap/1e6,'n gaps',len(gaps))
big=sorted(gaps,reverse=True)[:12]
print([(round(g/1e3,1),n) for g,n in big])
import collections
c=collections.Counter()
for g,n in gaps: c[n]+=g
print([(n,round(v/1e6,2)) for n,v in c.most_common(8)])

import csv, sys, collections
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ","")[:40], r.get("Stream_Id", r.get("Queue_Id","?"))))
rows.sort()
# take the last pass: find the last long gap (> 20 ms) separating passes
t_end=rows[-1][1]
# window: last 0.1 s of activity
cut=None
prev_end=rows[0][1]
starts=[]
for s,e,n,q in rows:
    if s-prev_end>5_000_000: starts.append(s)
    prev_end=max(prev_end,e)
t0=starts[-1] if starts else rows[0][0]
sel=[r for r in rows if r[0]>=t0]
wall=sel[-1][1]-sel[0][0]
# union busy
busy=0; cur_s,cur_e=sel[0][0],sel[0][1]
gaps=[]
for s,e,n,q in sel[1:]:
    if s>cur_e:
        busy+=cur_e-cur_s; gaps.append((s-cur_e,cur_e-t0,n)); cur_s,cur_e=s,e
    else: cur_e=max(cur_e,e)
busy+=cur_e-cur_s
print("last pass: wall %.2f ms, GPU busy (any kernel) %.2f ms, %d kernels"%(wall/1e6,busy/1e6,len(sel)))
tot=collections.Counter(); cnt=collections.Counter()
for s,e,n,q in sel: tot[n]+=e-s; cnt[n]+=1
for n,t in tot.most_common(12): print("  %-42s %8.2f ms %5d calls"%(n,t/1e6,cnt[n]))
gaps.sort(reverse=True)
print("largest idle gaps:")
for g,at,n in gaps[:12]: print("  %.2f ms at +%.2f ms before %s"%(g/1e6,at/1e6,n))

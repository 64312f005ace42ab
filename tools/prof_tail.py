import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last full iteration sequence before the final copies
idx=[i for i,r in enumerate(rows) if 'ransac_first' in r['Kernel_Name']]
end=idx[-1]
last=rows[end-14:end+1]
t0=int(last[0]['Start_Timestamp']); pe=None
for r in last:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    name=r['Kernel_Name'].replace('pclmi::(anonymous namespace)::','').split('(')[0][-40:]
    print("%-42s start %8.1f dur %7.1f gap %6.1f"%(name,(s-t0)/1e3,(e-s)/1e3,(s-pe)/1e3 if pe else 0)); pe=e

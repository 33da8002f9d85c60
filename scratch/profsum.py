import csv,glob,sys
d=sys.argv[1]; n=float(sys.argv[2])
f=glob.glob(d+'/*/*_kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('per step ms', tot/n/1e6)
for r in rows[:int(sys.argv[3]) if len(sys.argv)>3 else 50]:
    print(f"{float(r['TotalDurationNs'])/n/1e6:8.3f} ms/step {int(r['Calls'])/n:7.1f} calls {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:120]}")

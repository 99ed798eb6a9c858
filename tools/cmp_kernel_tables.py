"""Per-kernel differences of two rocprofv3 kernel tables of the training step (launches per step x average us).  usage: python tools/cmp_kernel_tables.py a.csv b.csv"""
import csv,sys
def load(f):
    rows=list(csv.DictReader(open(f)))
    st=[int(r["Calls"]) for r in rows if "adamw" in r["Name"]][0]
    return {r["Name"]:(int(r["Calls"])/st, float(r["AverageNs"])/1e3) for r in rows if int(r["Calls"])/st>=0.9}, st
a,_=load(sys.argv[1]); b,_=load(sys.argv[2])
ta=sum(n*t for n,t in a.values()); tb=sum(n*t for n,t in b.values())
print("busy us/step: 39=0 %.1f  39=1 %.1f" % (ta,tb))
for k in sorted(set(a)|set(b), key=lambda k:-(a.get(k,(0,0))[0]*a.get(k,(0,0))[1]+b.get(k,(0,0))[0]*b.get(k,(0,0))[1])):
    x=a.get(k,(0,0)); y=b.get(k,(0,0))
    if abs(x[0]*x[1]-y[0]*y[1])>8: print("%-95s %5.1f x %6.2f | %5.1f x %6.2f" % (k[:95], x[0],x[1],y[0],y[1]))

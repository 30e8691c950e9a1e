import csv,sys,collections,re
acc=collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"].replace("(anonymous namespace)::","").replace("void ","")
    fam=re.split(r"[<(]",n)[0]
    acc[fam][0]+=int(r["Calls"]); acc[fam][1]+=float(r["TotalDurationNs"])
steps=float(sys.argv[2])
tot=0
for k,v in sorted(acc.items(), key=lambda x:-x[1][1])[:22]:
    print(f"   {k[:40]:40s} {v[0]/steps:7.1f} launches/step {v[1]/steps/1e6:8.3f} ms/step")
    tot+=v[1]
print("   total", round(sum(v[1] for v in acc.values())/steps/1e6,3))

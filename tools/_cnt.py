import json,sys
v=[]
for l in sys.stdin:
    if l.startswith("{"):
        v.append(json.loads(l)["ms_per_step"])
print(sys.argv[1], " ".join("%.3f"%x for x in v))

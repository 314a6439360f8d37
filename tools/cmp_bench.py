import json,sys
a=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); b=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("value", a["value"], b["value"], "ms", a["ms_per_step"], b["ms_per_step"])
for k in a["roofline_all"]:
    x=a["roofline_all"][k]; y=b["roofline_all"].get(k,{})
    print(f"{k:16s} {x['ms_per_step']:8.3f} {y.get('ms_per_step',0):8.3f}  {100*(y.get('ms_per_step',0)/x['ms_per_step']-1):+6.1f}%")

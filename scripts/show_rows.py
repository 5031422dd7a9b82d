import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    print(k, v.get("ms"), json.dumps(v.get("kernels_ms")), v.get("other_ms"))

import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]
print(d["value"], "pass_ms", d["roofline"]["kernel_ms"], {n:v["avg_ms"] for n,v in k.items()})

import json,sys
for ln in sys.stdin:
    if ln.startswith("{"):
        d=json.loads(ln); print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],3), "r200k", round(d["r200k"]["value"],1))

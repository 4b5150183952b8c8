import json,sys
d=json.load(open(sys.argv[1]))
print("headline ms", round(d["ms_per_step"],3), "median", round(d["median_ms_per_step"],3), "frac", round(d["roofline"]["frac"],4), "value", round(d["value"]))
print("single", d["single_shot_ms"]["qc_qft_then_maxprob_ms"])
for k,v in d["configs"].items(): print(k, v.get("ms_per_step"), v.get("roofline",{}).get("frac"), v.get("error"))
print("ladder", d["ladder_base"].get("ms_per_step"))

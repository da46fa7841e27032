import json,sys
d=json.load(open(sys.argv[1]))
s=d.get("summary",{})
print("value", d["value"], "ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"])
for k,v in s.items(): print("  ",k, v)
i=d["secondary"]["icp_verification"]
for est in ("point_to_point","point_to_plane"):
    for path in ("from_store","host_buffers"):
        print("  icp",est,path, round(i[est][path]["ms_per_query"],3))
print("  icp ref settings", round(i["point_to_point_reference_settings"]["ms_per_query"],3))

"""Appends the traffic entries tools/profile.sh left under gpurun_out/prof_<tag>/traffic_entry.json to profiles/traffic.json (what bench.py's
`roofline.traffic` is looked up in, by the hash of the scoring sources) and copies every summary to profiles/<tag>_summary.txt:
python tools/merge_traffic.py <tag> [<tag> ...]"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(ROOT, "profiles", "traffic.json")
doc = json.load(open(path))
for tag in sys.argv[1:]:
    d = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    e = os.path.join(d, "traffic_entry.json")
    if os.path.exists(e):
        entry = json.load(open(e))
        doc["entries"] = [x for x in doc["entries"] if not (x.get("round") == entry["round"])] + [entry]
        print("added", entry["round"], entry["workload"], entry["read_bytes"] + entry["write_bytes"])
    else:
        print("no traffic entry for", tag)
    if os.path.exists(os.path.join(d, "summary.txt")):
        shutil.copy(os.path.join(d, "summary.txt"), os.path.join(ROOT, "profiles", tag + "_summary.txt"))
json.dump(doc, open(path, "w"), indent=1)

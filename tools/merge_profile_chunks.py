"""Merge the trimmed outputs of several tools/profile_round.sh calls (gpurun_out/prof_<tag>_small) into profiles/<round>/rocprof/:
case directories copied (later chunks win), summary.json merged.   usage: python tools/merge_profile_chunks.py r03 r03a r03b r03c ..."""
import json
import os
import shutil
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, tags = sys.argv[1], sys.argv[2:]
dst = os.path.join(root, "profiles", rnd, "rocprof")
os.makedirs(dst, exist_ok=True)
summ = {}
sp = os.path.join(dst, "summary.json")
if os.path.exists(sp):
    summ = json.load(open(sp))
for tag in tags:
    src = os.path.join(root, "gpurun_out", f"prof_{tag}_small")
    part = json.load(open(os.path.join(src, "summary.json")))
    for case in sorted(os.listdir(src)):
        p = os.path.join(src, case)
        if os.path.isdir(p) and case in part:
            shutil.rmtree(os.path.join(dst, case), ignore_errors=True)
            shutil.copytree(p, os.path.join(dst, case))
            summ[case] = part[case]
json.dump(summ, open(sp, "w"), indent=1, sort_keys=True)
print(len(summ), "cases:", " ".join(sorted(summ)))

#!/usr/bin/env python3
"""Sums rocprofv3 --pmc CSV output per kernel name.  usage: pmc_summary.py <dir> [<dir>...]"""
import csv, glob, sys, collections, re

def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.match(r"(?:void )?([A-Za-z_0-9]+(?:<[^>]*>)?)", n)
    return m.group(1) if m else n[:40]

for d in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k].add(r["Dispatch_Id"])
    print("==", d)
    for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
        if k.startswith("__amd") or "elementwise" in k: continue
        print("%-28s calls=%d " % (k, len(calls[k])) + " ".join("%s=%.4g" % kv for kv in sorted(acc[k].items())))

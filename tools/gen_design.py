#!/usr/bin/env python3
"""Rewrites the generated blocks of DESIGN.md from the files under profiles/ -- numbers in DESIGN.md are never typed by hand:

    python tools/gen_design.py [tag]

A block is everything between `<!-- BEGIN name -->` and `<!-- END name -->`; `name` maps to profiles/<tag>_<file>.md (its heading line
dropped).  Run after tools/parity_table.py --design, tools/perf_table.py."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
BLOCKS = {"parity_table": "parity_table_design.md", "warped_table": "warped_table_design.md", "perf_table": "perf_table.md"}
path = os.path.join(ROOT, "DESIGN.md")
text = open(path).read()
for name, fname in BLOCKS.items():
    src = os.path.join(ROOT, "profiles", "%s_%s" % (tag, fname))
    if not os.path.exists(src):
        print("skipped %s: %s is missing" % (name, src))
        continue
    body = open(src).read().split("\n", 2)[2] if open(src).read().startswith("#") else open(src).read()
    pat = re.compile(r"(<!-- BEGIN %s -->\n).*?(<!-- END %s -->)" % (name, name), re.S)
    if not pat.search(text):
        print("DESIGN.md has no block %s" % name)
        continue
    text = pat.sub(lambda m: m.group(1) + "(generated from profiles/%s_%s by tools/gen_design.py)\n\n" % (tag, fname) + body.strip("\n") + "\n" + m.group(2), text)
    print("block %s <- %s" % (name, os.path.relpath(src, ROOT)))
open(path, "w").write(text)

#!/usr/bin/env python3
"""Checks every `file:line` / `file:line-line` citation of the reference in this repo's sources and docs: the cited file must exist
under /root/reference and the line numbers must lie inside it.  (Build-container tool: needs /root/reference.)

  python tools/check_citations.py [-v]
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("GY_REFERENCE_DIR", "/root/reference")
EXTS = (".h", ".hpp", ".hip", ".c", ".cc", ".py", ".md", ".sh")
SKIP_DIRS = {".git", "gpurun_out", "__pycache__", "_ref", "golden"}
SKIP_FILES = {"SURVEY.md", "PAPERS.md", "SNIPPETS.md", "BASELINE.md"}
PAT = re.compile(r"((?:[\w\-]+/)*[\w\-]+\.(?:h|cc|c|hpp|js|go))\s*:\s*(\d+)(?:\s*-\s*(\d+))?")


def index_reference():
    by_base, by_path = {}, {}
    for d, _, files in os.walk(REF):
        for f in files:
            p = os.path.join(d, f)
            rel = os.path.relpath(p, REF)
            by_path[rel] = p
            by_base.setdefault(f, []).append(p)
    return by_base, by_path


def nlines(path, cache={}):
    if path not in cache:
        with open(path, "rb") as fh:
            cache[path] = fh.read().count(b"\n") + 1
    return cache[path]


def main():
    verbose = "-v" in sys.argv
    if not os.path.isdir(REF):
        print("no reference tree at", REF)
        return 0
    by_base, by_path = index_reference()
    own = set()
    for d, dirs, files in os.walk(ROOT):
        dirs[:] = [x for x in dirs if x not in SKIP_DIRS]
        for f in files:
            own.add(f)
    bad = total = 0
    for d, dirs, files in os.walk(ROOT):
        dirs[:] = [x for x in dirs if x not in SKIP_DIRS]
        for f in files:
            if not f.endswith(EXTS) or f in SKIP_FILES:
                continue
            p = os.path.join(d, f)
            text = open(p, errors="replace").read()
            for m in PAT.finditer(text):
                name, lo, hi = m.group(1), int(m.group(2)), int(m.group(3) or m.group(2))
                base = os.path.basename(name)
                cands = [by_path[name]] if name in by_path else [c for c in by_base.get(base, []) if c.endswith("/" + name) or "/" not in name]
                if not cands:
                    if base in own or base.startswith(("gys_", "gy_oracle", "gysketch", "ref_glue", "test_")):
                        continue  # a citation of this repo's own files
                    print(f"{os.path.relpath(p, ROOT)}: cites {name}:{lo} -- no such file in the reference")
                    bad += 1
                    continue
                total += 1
                ok = any(lo >= 1 and hi >= lo and hi <= nlines(c) for c in cands)
                if not ok:
                    print(f"{os.path.relpath(p, ROOT)}: cites {name}:{lo}-{hi} -- outside the file ({', '.join(str(nlines(c)) for c in cands)} lines)")
                    bad += 1
                elif verbose:
                    print(f"ok {os.path.relpath(p, ROOT)}: {name}:{lo}-{hi}")
    print(f"{total} citations checked, {bad} problems")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

"""Re-wrap the prose of a Markdown file to <= 120 columns: paragraphs and list items are re-flowed with a
hanging indent; headings, tables, fenced code, indented code and HTML are left alone.  No words change.

    python scripts/wrap_md.py DESIGN.md [width]
"""
import re
import sys
import textwrap

path = sys.argv[1]
W = int(sys.argv[2]) if len(sys.argv) > 2 else 120
lines = open(path).read().split("\n")
out, para, fence = [], [], False
BUL = re.compile(r"^(\s*)([*+-]|\d+\.)\s+")


def flush():
    if not para:
        return
    first = para[0]
    m = BUL.match(first)
    if m:
        ind0 = m.group(0)
        ind = " " * len(ind0)
        body = first[len(ind0):]
    else:
        lead = len(first) - len(first.lstrip(" "))
        ind0 = ind = " " * lead
        body = first.lstrip(" ")
    text = " ".join([body.strip()] + [l.strip() for l in para[1:]])
    # keep double spaces after sentence ends as they are: textwrap collapses nothing unless asked
    # greedy fill measured in UTF-8 BYTES (`awk length` in the C locale counts bytes: x, ->, us ... are 2-3 each)
    wrapped, cur = [], ind0
    for word in text.split(" "):
        if word == "" and cur.strip():
            cur += " "                     # a double space after a sentence end survives inside a line
            continue
        cand = cur + ("" if cur == ind0 or cur.endswith(" ") else " ") + word if cur.strip() else cur + word
        if len(cand.encode()) > W and cur.strip():
            wrapped.append(cur.rstrip())
            cur = ind + word
        else:
            cur = cand
    if cur.strip():
        wrapped.append(cur.rstrip())
    out.extend(wrapped if wrapped else [first])
    para.clear()


for ln in lines:
    s = ln.strip()
    if s.startswith("```"):
        flush()
        fence = not fence
        out.append(ln)
        continue
    if fence or not s or s.startswith("#") or s.startswith("|") or s.startswith("<") or ln.startswith("    ") and not para:
        flush()
        out.append(ln)
        continue
    if BUL.match(ln):
        flush()
        para.append(ln)
        continue
    para.append(ln)
flush()
open(path, "w").write("\n".join(out))
long_ = [i + 1 for i, l in enumerate(out) if len(l.encode()) > W and not l.strip().startswith("|")]
print("%s: %d lines, %d prose lines still > %d columns (unbreakable tokens)" % (path, len(out), len(long_), W), long_[:10])

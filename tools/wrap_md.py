#!/usr/bin/env python3
"""Wrap the prose of a markdown file to a column limit: paragraphs and list items (continuation lines indented under the item's
text); headings, tables, fenced code and indented code blocks are left as they are.   python tools/wrap_md.py FILE [width=120]"""
import re
import sys
import textwrap

path = sys.argv[1]
width = int(sys.argv[2]) if len(sys.argv) > 2 else 120
out, para, fence = [], [], False


def flush():
    global para
    if not para:
        return
    first = para[0]
    m = re.match(r"^(\s*)([*+-]|\d+[.)])\s+", first)
    if m:
        indent0, indent1 = m.group(0), " " * len(m.group(0))
        text = " ".join([first[len(m.group(0)):].strip()] + [l.strip() for l in para[1:]])
    else:
        lead = re.match(r"^\s*", first).group(0)
        indent0 = indent1 = lead
        text = " ".join(l.strip() for l in para)
    out.extend(textwrap.wrap(text, width=width, initial_indent=indent0, subsequent_indent=indent1, break_long_words=False,
                             break_on_hyphens=False) or [""])
    para = []


for line in open(path).read().split("\n"):
    if line.lstrip().startswith("```"):
        flush(); fence = not fence; out.append(line); continue
    if fence or line.startswith("#") or line.lstrip().startswith("|") or line.strip() == "" or line.startswith("    ") and not para:
        flush(); out.append(line); continue
    if re.match(r"^\s*([*+-]|\d+[.)])\s+", line) and para:
        flush()
    para.append(line)
flush()
open(path, "w").write("\n".join(out))

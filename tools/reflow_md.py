#!/usr/bin/env python3
"""Re-flow a markdown file to lines of at most WIDTH columns (default 118): paragraphs and list items are wrapped with their
indentation kept; a table any of whose rows is wider than WIDTH becomes a list -- one item per row, "**first cell**" then one
sub-item "header: cell" per further non-empty cell -- because a table cell cannot be wrapped; code blocks, headings and tables
that fit are left alone.  usage: reflow_md.py FILE [WIDTH]   (rewrites FILE in place)"""
import re
import sys
import textwrap

path = sys.argv[1]
W = int(sys.argv[2]) if len(sys.argv) > 2 else 118
src = open(path).read().split("\n")
out = []


def wrap(text, first, rest):
    return textwrap.wrap(text, width=W, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False) or [first.rstrip()]


def cells(row):
    row = row.strip()
    if row.startswith("|"):
        row = row[1:]
    if row.endswith("|"):
        row = row[:-1]
    return [c.strip() for c in re.split(r"(?<!\\)\|", row)]


i = 0
while i < len(src):
    line = src[i]
    if line.lstrip().startswith("```"):                      # code block: verbatim
        out.append(line); i += 1
        while i < len(src) and not src[i].lstrip().startswith("```"):
            out.append(src[i]); i += 1
        if i < len(src):
            out.append(src[i]); i += 1
        continue
    if line.lstrip().startswith("|") and i + 1 < len(src) and re.match(r"^\s*\|?\s*:?-{3,}", src[i + 1]):   # a table
        indent = line[:len(line) - len(line.lstrip())]
        j = i
        rows = []
        while j < len(src) and src[j].lstrip().startswith("|"):
            rows.append(src[j]); j += 1
        if max(len(r) for r in rows) <= W:
            out.extend(rows)
        else:
            head = cells(rows[0])
            for r in rows[2:]:
                c = cells(r)
                out.extend(wrap(f"**{c[0]}**" if c and c[0] else "**-**", indent + "- ", indent + "  "))
                for h, v in zip(head[1:], c[1:]):
                    if v:
                        out.extend(wrap((f"*{h}*: " if h else "") + v, indent + "  - ", indent + "    "))
        i = j
        continue
    if not line.strip() or line.lstrip().startswith("#") or len(line) <= W and not (i + 1 < len(src) and src[i + 1].strip() and not re.match(r"^\s*([-*+]|\d+\.)\s|^\s*#|^\s*\||^\s*```", src[i + 1]) and len(line) > W - 25):
        # blank, heading, or a short line that is not the start of a paragraph worth re-flowing
        if len(line) <= W:
            out.append(line); i += 1
            continue
    # a paragraph or list item: gather its continuation lines (same block: no blank line, no new item, no table, no heading)
    m = re.match(r"^(\s*)((?:[-*+]|\d+[.)])\s+)?(.*)$", line)
    indent, bullet, text = m.group(1), m.group(2) or "", m.group(3)
    j = i + 1
    while j < len(src) and src[j].strip() and not re.match(r"^\s*([-*+]|\d+[.)])\s|^\s*#|^\s*\||^\s*```", src[j]):
        text += " " + src[j].strip(); j += 1
    out.extend(wrap(text, indent + bullet, indent + " " * len(bullet)))
    i = j
open(path, "w").write("\n".join(out))
print(path, "->", len(out), "lines; longest", max(len(l) for l in out))

#!/usr/bin/env python3
"""Reflow a Markdown file to a column limit (default 118): paragraphs and list items are re-wrapped with their indent kept,
fenced code and headings are left alone, and a table with a row wider than the limit is rewritten as a list (one item per
row, one sub-item per column, headed by the column's title) -- Markdown tables cannot be wrapped.

usage: scripts/reflow_md.py FILE [--width N] [--out FILE]   (here, CPU only; used to split DESIGN.md in round 5)"""
import re
import sys
import textwrap


def wrap(text, first, rest, width):
    # the limit is in BYTES of UTF-8 (arrows and dashes are three each): narrow the wrap until every line fits
    w = width
    while True:
        t = textwrap.fill(text, width=w, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False)
        if w <= 60 or all(len(l.encode()) <= width or " " not in l.strip() for l in t.split("\n")):
            return t
        w -= 4


def split_row(line):
    cells = [c.strip() for c in re.split(r"(?<!\\)\|", line.strip())]
    if cells and cells[0] == "":
        cells = cells[1:]
    if cells and cells[-1] == "":
        cells = cells[:-1]
    return cells


def table_as_list(rows, width):
    head = split_row(rows[0])
    out = []
    for r in rows[2:]:
        cells = split_row(r)
        if not cells:
            continue
        out.append(wrap(f"**{cells[0]}**" if cells[0] else "**-**", "- ", "  ", width))
        for h, c in zip(head[1:], cells[1:]):
            if c:
                out.append(wrap(f"*{h}*: {c}" if h else c, "  - ", "    ", width))
    return out


def reflow(lines, width):
    out, i, n = [], 0, len(lines)
    while i < n:
        ln = lines[i].rstrip("\n")
        if ln.lstrip().startswith("```"):
            out.append(ln)
            i += 1
            while i < n and not lines[i].lstrip().startswith("```"):
                out.append(lines[i].rstrip("\n"))
                i += 1
            if i < n:
                out.append(lines[i].rstrip("\n"))
                i += 1
            continue
        if ln.startswith("|"):
            rows = []
            while i < n and lines[i].startswith("|"):
                rows.append(lines[i].rstrip("\n"))
                i += 1
            if max(len(r) for r in rows) > width + 2 and len(rows) >= 2 and re.match(r"^\|[\s:|-]+\|?\s*$", rows[1]):
                out.extend(table_as_list(rows, width))
            else:
                out.extend(rows)
            continue
        if ln.strip() == "" or ln.startswith("#") or re.match(r"^\s*(---+|===+)\s*$", ln):
            out.append(ln)
            i += 1
            continue
        # a paragraph or a list item: first line decides the indents; continuation lines are indented text that does not start a new item
        m = re.match(r"^(\s*)((?:[-*+]|\d+[.)])\s+)?(.*)$", ln)
        ind, bullet, body = m.group(1), m.group(2) or "", m.group(3)
        first = ind + bullet
        rest = ind + " " * len(bullet)
        parts = [body]
        i += 1
        while i < n:
            nx = lines[i].rstrip("\n")
            if nx.strip() == "" or nx.startswith("#") or nx.startswith("|") or nx.lstrip().startswith("```"):
                break
            if re.match(r"^\s*(?:[-*+]|\d+[.)])\s+", nx):
                break
            parts.append(nx.strip())
            i += 1
        out.append(wrap(" ".join(parts), first, rest, width))
    return out


def main():
    args = sys.argv[1:]
    width, outp = 118, None
    if "--width" in args:
        k = args.index("--width")
        width = int(args[k + 1])
        del args[k:k + 2]
    if "--out" in args:
        k = args.index("--out")
        outp = args[k + 1]
        del args[k:k + 2]
    src = args[0]
    res = reflow(open(src).read().split("\n"), width)
    text = "\n".join(res)
    if not text.endswith("\n"):
        text += "\n"
    open(outp or src, "w").write(text)
    wide = [l for l in res if len(l.encode()) > width + 2]
    print(f"{outp or src}: {len(res)} lines, {len(wide)} still wider than {width} (unbreakable tokens, narrow tables, code)")


if __name__ == "__main__":
    main()

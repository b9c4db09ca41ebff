#!/usr/bin/env python3
"""tools/md_reflow.py FILE... -- re-flow markdown files in place to lines of at most 118 characters.

Paragraphs and list items are re-wrapped (list markers and their hanging indent kept); headings, fenced code and tables that
fit stay as they are.  A table with a row that does not fit is rewritten as a list: one item per row, the first cell in bold,
the other cells as `header: cell` sentences underneath.  `--check` only reports lines over 120 characters."""
import re, sys, textwrap

WIDTH = 118
LIST = re.compile(r"^(\s*)([*+-]|\d+[.)])(\s+)")


def wrap(text, first, rest):
    return textwrap.wrap(" ".join(text.split()), WIDTH, initial_indent=first, subsequent_indent=rest,
                         break_long_words=False, break_on_hyphens=False) or [first.rstrip()]


def cells(row):
    row = row.strip()
    if row.startswith("|"): row = row[1:]
    if row.endswith("|"): row = row[:-1]
    out, cur, code = [], "", False
    for ch in row:
        if ch == "`": code = not code
        if ch == "|" and not code and not cur.endswith("\\"):
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    out.append(cur.strip())
    return out


def table_to_list(rows):
    head = cells(rows[0]); body = [cells(r) for r in rows[2:]]
    out = []
    for r in body:
        if not any(r): continue
        out += wrap(f"**{r[0]}**" if r[0] else "**-**", "* ", "  ")
        for h, c in zip(head[1:], r[1:]):
            if c and c not in ("–", "-"):
                out += wrap(f"{h}: {c}" if h else c, "  - ", "    ")
    return out


def reflow(lines):
    out, i, n = [], 0, len(lines)
    while i < n:
        l = lines[i].rstrip("\n")
        if l.startswith("```"):
            out.append(l); i += 1
            while i < n and not lines[i].startswith("```"): out.append(lines[i].rstrip("\n")); i += 1
            if i < n: out.append(lines[i].rstrip("\n")); i += 1
            continue
        if l.startswith("|"):
            rows = []
            while i < n and lines[i].startswith("|"): rows.append(lines[i].rstrip("\n")); i += 1
            if max(len(r) for r in rows) <= 120 or len(rows) < 3: out += rows
            else: out += table_to_list(rows)
            continue
        if not l.strip() or l.startswith("#") or l.startswith("<!--"):
            out.append(l); i += 1
            continue
        # a paragraph or one list item with its continuation lines
        m = LIST.match(l)
        first = m.group(0) if m else re.match(r"^\s*", l).group(0)
        rest = " " * len(first) if m else first
        text = l[len(first):]
        i += 1
        while i < n:
            nl = lines[i].rstrip("\n")
            if not nl.strip() or nl.startswith("#") or nl.startswith("|") or nl.startswith("```") or LIST.match(nl): break
            text += " " + nl.strip(); i += 1
        out += wrap(text, first, rest)
    return out


if __name__ == "__main__":
    check = "--check" in sys.argv
    bad = 0
    for p in [a for a in sys.argv[1:] if not a.startswith("--")]:
        src = open(p).read().split("\n")
        if check:
            for k, l in enumerate(src):
                if len(l) > 120: bad += 1; print(f"{p}:{k + 1}: {len(l)} characters")
            continue
        res = reflow(src)
        while res and not res[-1]: res.pop()
        open(p, "w").write("\n".join(res) + "\n")
    sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Wrap the prose of a markdown file at WIDTH columns (default 150) without touching tables, code fences, headings or display lines that are short already:
paragraphs and list items are re-flowed (list items keep a hanging indent).  wrap_md.py <file> [width]"""
import re
import sys
import textwrap


def main(path, width=150):
    lines = open(path).read().split("\n")
    out, para, fence = [], [], False

    def flush():
        if not para:
            return
        first = para[0]
        m = re.match(r"^(\s*)([-*]|\d+\.)\s+", first)
        if m:
            lead = m.group(0)
            hang = " " * len(lead)
            text = " ".join([first[len(lead):].strip()] + [p.strip() for p in para[1:]])
            out.extend(textwrap.wrap(text, width=width, initial_indent=lead, subsequent_indent=hang, break_long_words=False, break_on_hyphens=False))
        else:
            indent = re.match(r"^\s*", first).group(0)
            text = " ".join(p.strip() for p in para)
            out.extend(textwrap.wrap(text, width=width, initial_indent=indent, subsequent_indent=indent, break_long_words=False, break_on_hyphens=False))
        para.clear()

    for ln in lines:
        if ln.strip().startswith("```"):
            flush()
            fence = not fence
            out.append(ln)
            continue
        if fence or ln.startswith("|") or ln.startswith("#") or ln.strip() == "" or ln.startswith("    ") and not para:
            flush()
            out.append(ln)
            continue
        if re.match(r"^\s*([-*]|\d+\.)\s+", ln):  # a new list item ends the previous paragraph / item
            flush()
        para.append(ln)
    flush()
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 150)

#!/usr/bin/env python3
"""Re-wrap the prose of a Markdown file at <= 160 columns (round-5 verdict, hygiene: DESIGN.md had 17 lines above 300 characters).  Tables, fenced and indented code,
headings and lines that already fit are left alone; a bullet's continuation lines hang under its text.  usage: tools/wrap_md.py FILE... [--width 160] [--check]"""
import re
import sys
import textwrap


def wrap_file(path, width, check):
    out, fence, changed = [], False, 0
    for line in open(path).read().split("\n"):
        if line.lstrip().startswith("```"):
            fence = not fence
        if fence or len(line) <= width or line.lstrip().startswith("|") or line.startswith("    ") and not re.match(r"\s*([*+-]|\d+\.)\s", line) or line.startswith("#"):
            out.append(line)
            continue
        m = re.match(r"(\s*(?:[*+-]|\d+\.)\s+|\s*)", line)
        first = m.group(1)
        hang = " " * len(first)
        body = line[len(first):]
        wrapped = textwrap.wrap(body, width=width - len(first), break_long_words=False, break_on_hyphens=False)
        out.append(first + wrapped[0])
        out.extend(hang + w for w in wrapped[1:])
        changed += 1
    if check:
        long_lines = [i + 1 for i, l in enumerate(out) if len(l) > width and not l.lstrip().startswith("|")]
        print("%s: %d prose lines above %d columns" % (path, len(long_lines), width))
        return len(long_lines)
    open(path, "w").write("\n".join(out))
    print("%s: %d lines re-wrapped" % (path, changed))
    return 0


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    width = int(sys.argv[sys.argv.index("--width") + 1]) if "--width" in sys.argv else 160
    if "--width" in sys.argv:
        args.remove(str(width))
    bad = sum(wrap_file(p, width, "--check" in sys.argv) for p in args)
    sys.exit(1 if bad and "--check" in sys.argv else 0)


if __name__ == "__main__":
    main()

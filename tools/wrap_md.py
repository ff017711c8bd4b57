#!/usr/bin/env python3
"""Re-wrap the prose of a Markdown file at <= 160 columns (round-5 verdict, hygiene: DESIGN.md had 17 lines above 300 characters).  Tables, fenced and indented code,
headings and lines that already fit are left alone; a bullet's continuation lines hang under its text.  usage: tools/wrap_md.py FILE... [--width 160] [--check]; tools/wrap_md.py --reflow FILE... [--width 150] joins and re-wraps whole paragraphs"""
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


def reflow_file(path, width):
    """Join the lines of every prose paragraph / list item and wrap them again (a file wrapped twice at different widths carries orphan half-lines).
    Tables, fenced and indented code, headings and blank lines separate blocks and are kept as they are."""
    src = open(path).read().split("\n")
    out, fence = [], False
    block = None                                        # [first prefix, hang, [text parts], is_item]
    marker = re.compile(r"(\s*(?:[*+-]|\d+\.)\s+)(.*)$")

    def flush():
        nonlocal block
        if block:
            first, hang, parts, _ = block
            body = " ".join(x.strip() for x in parts if x.strip())
            # keep the two-space sentence gaps of the source out of the way of textwrap (it would break inside them unevenly): normalise to one, as Markdown renders it
            wrapped = textwrap.wrap(body, width=width - len(first), break_long_words=False, break_on_hyphens=False) or [""]
            out.append(first + wrapped[0])
            out.extend(hang + w for w in wrapped[1:])
            block = None

    for line in src:
        if line.lstrip().startswith("```"):
            flush(); fence = not fence; out.append(line); continue
        if fence:
            out.append(line); continue
        if not line.strip() or line.startswith("#") or line.lstrip().startswith("|") or re.match(r"\s*(---+|===+)\s*$", line):
            flush(); out.append(line); continue
        m = marker.match(line)
        if m and (block is None or len(m.group(1)) - len(m.group(1).lstrip()) <= 8):
            flush(); first = m.group(1); block = [first, " " * len(first), [m.group(2)], True]; continue
        indent = len(line) - len(line.lstrip())
        if block is None:
            if indent >= 4:                              # indented code / a formula block
                out.append(line); continue
            block = [" " * indent, " " * indent, [line], False]; continue
        if block[3] or indent < 4:                       # a list item's continuation (any indentation), or the paragraph's next line
            block[2].append(line); continue
        flush(); out.append(line)
    flush()
    before = re.sub(r"\s+", "", "\n".join(src)); after = re.sub(r"\s+", "", "\n".join(out))
    assert before == after, "reflow changed more than white space in %s" % path
    open(path, "w").write("\n".join(out))
    print("%s: reflowed at %d columns, %d -> %d lines" % (path, width, len(src), len(out)))
    return 0


def main():
    if "--reflow" in sys.argv:
        args = [a for a in sys.argv[1:] if not a.startswith("--")]
        width = int(sys.argv[sys.argv.index("--width") + 1]) if "--width" in sys.argv else 150
        if "--width" in sys.argv:
            args.remove(str(width))
        for p in args:
            reflow_file(p, width)
        return
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    width = int(sys.argv[sys.argv.index("--width") + 1]) if "--width" in sys.argv else 160
    if "--width" in sys.argv:
        args.remove(str(width))
    bad = sum(wrap_file(p, width, "--check" in sys.argv) for p in args)
    sys.exit(1 if bad and "--check" in sys.argv else 0)


if __name__ == "__main__":
    main()

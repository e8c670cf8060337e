"""Re-flow a markdown file's paragraphs and bullets to a column limit (tables, headings, fenced code and indented blocks are left alone).
usage: python tools/wrap_md.py FILE [WIDTH=120]"""
import sys


def wrap(text, width, initial_indent="", subsequent_indent=""):
    """Greedy wrap on spaces; the limit counts UTF-8 BYTES (what `awk 'length > N'` and most line-length checks see)."""
    lines, cur, fresh = [], initial_indent, True
    for word in text.split():
        cand = cur + ("" if fresh else " ") + word
        if len(cand.encode()) > width and not fresh:
            lines.append(cur)
            cur = subsequent_indent + word
        else:
            cur = cand
        fresh = False
    if not fresh:
        lines.append(cur)
    return lines

path = sys.argv[1]
width = int(sys.argv[2]) if len(sys.argv) > 2 else 120
out, block, fenced = [], [], False


def flush():
    if not block:
        return
    first = block[0]
    if first.startswith("* ") or first.startswith("- "):
        text = " ".join(l.strip() for l in block)[2:]
        out.extend(wrap(text, width, first[:2], "  "))
    else:
        out.extend(wrap(" ".join(l.strip() for l in block), width))
    block.clear()


for line in open(path).read().split("\n"):
    if line.startswith("```"):
        flush(); fenced = not fenced; out.append(line); continue
    if fenced or not line.strip() or line.startswith("#") or line.startswith("|") or line.startswith("    "):
        flush(); out.append(line); continue
    if (line.startswith("* ") or line.startswith("- ")) and block:
        flush()
    if block and not (block[0].startswith("* ") or block[0].startswith("- ")) and line.startswith("  "):
        flush()
    block.append(line)
flush()
open(path, "w").write("\n".join(out))

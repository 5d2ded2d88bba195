"""Hard-wraps the prose of a markdown file at 120 columns (tests/test_docs_format.py): paragraphs and list items are re-flowed with
their indentation, tables, headings, code blocks and blank lines are left alone.  `python3 profiles/wrap_docs.py FILE...`"""
import re
import sys
import textwrap

WIDTH = 120


def wrap(path):
    out, block, code = [], [], False

    def flush():
        if not block:
            return
        first = block[0]
        m = re.match(r"^(\s*)((?:[-*]|\d+\.)\s+)?", first)
        indent, bullet = m.group(1), m.group(2) or ""
        text = " ".join(line.strip() for line in block)
        if bullet:
            text = text[len(bullet.strip()):].lstrip() if text.startswith(bullet.strip()) else text
        body = textwrap.fill(text, WIDTH, initial_indent=indent + bullet, subsequent_indent=indent + " " * len(bullet),
                             break_long_words=False, break_on_hyphens=False)
        out.extend(body.split("\n"))
        block.clear()

    for line in open(path, encoding="utf8").read().split("\n"):
        stripped = line.strip()
        if stripped.startswith("```"):
            flush()
            code = not code
            out.append(line)
            continue
        if code or stripped.startswith("|") or stripped.startswith("#") or stripped == "" or stripped.startswith("{"):
            flush()
            out.append(line)
            continue
        if re.match(r"^\s*(?:[-*]|\d+\.)\s+", line) and block:
            flush()
        # a line indented differently from the block's continuation starts a new block
        if block and not re.match(r"^\s*(?:[-*]|\d+\.)\s+", block[0]) and (len(line) - len(line.lstrip())) != (len(block[0]) - len(block[0].lstrip())):
            flush()
        block.append(line)
    flush()
    open(path, "w", encoding="utf8").write("\n".join(out))


if __name__ == "__main__":
    for p in sys.argv[1:]:
        wrap(p)

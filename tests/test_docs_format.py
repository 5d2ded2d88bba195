"""The evidence stays readable: DESIGN.md and the round's README hard-wrapped at 120 columns (tables included), the other
documents outside tables and code blocks."""
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def long_lines(path, tables_too):
    out, code = [], False
    for i, line in enumerate(open(os.path.join(ROOT, path), encoding="utf8"), 1):
        line = line.rstrip("\n")
        if line.lstrip().startswith("```"):
            code = not code
            continue
        if code or (not tables_too and line.startswith("|")):
            continue
        if len(line) > 120:
            out.append((i, len(line)))
    return out


@pytest.mark.parametrize("path,tables_too", [("DESIGN.md", True), ("profiles/r05/README.md", True), ("profiles/r04/README.md", True), ("profiles/r03/README.md", False),
                                             ("profiles/experiments/README.md", False), ("INTEGRATION.md", False)])
def test_documents_are_wrapped(path, tables_too):
    assert long_lines(path, tables_too) == []

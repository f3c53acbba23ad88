#!/usr/bin/env python3
"""INTEGRATION.md carries EXCERPTS of the compiled binding oracle/hip_binding.f90, not a copy of it: every excerpt sits
between `<!-- BEGIN excerpt: <first line> ... <last line> -->` and `<!-- END excerpt -->` and is refreshed from the file here
(from the first line that starts with <first line> through the next line that starts with <last line>);
tests/test_cabi_cpu.py checks that every fenced Fortran block of the document is verbatim text of the file."""
import os
import re

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(root, "oracle", "hip_binding.f90")).read().splitlines()
p = os.path.join(root, "INTEGRATION.md")
doc = open(p).read()


def excerpt(first, last):
    i = next(k for k, ln in enumerate(src) if ln.startswith(first))
    j = next(k for k in range(i, len(src)) if src[k].startswith(last))
    return "\n".join(src[i:j + 1])


def repl(m):
    first, last = m.group(1), m.group(2)
    return f"<!-- BEGIN excerpt: {first} ... {last} -->\n```fortran\n{excerpt(first, last)}\n```\n<!-- END excerpt -->"


doc, n = re.subn(r"<!-- BEGIN excerpt: (.*?) \.\.\. (.*?) -->.*?<!-- END excerpt -->", repl, doc, flags=re.S)
open(p, "w").write(doc)
print(f"INTEGRATION.md: {n} excerpts of oracle/hip_binding.f90 refreshed")

#!/usr/bin/env python3
"""Copy oracle/hip_binding.f90 verbatim into INTEGRATION.md between the BEGIN/END markers
(tests/test_cabi_cpu.py checks that the document carries the compiled file, not a sketch)."""
import os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(root, "oracle", "hip_binding.f90")).read()
p = os.path.join(root, "INTEGRATION.md")
doc = open(p).read()
b, e = "<!-- BEGIN oracle/hip_binding.f90 -->", "<!-- END oracle/hip_binding.f90 -->"
i, j = doc.index(b) + len(b), doc.index(e)
doc = doc[:i] + "\n```fortran\n" + src.strip() + "\n```\n" + doc[j:]
open(p, "w").write(doc)
print("INTEGRATION.md synced with oracle/hip_binding.f90")

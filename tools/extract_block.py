"""Writes the cpp block of one INTEGRATION.md section to a file, so that a harness under oracle/ compiles and LINKS exactly the text the
document shows a maintainer.   python tools/extract_block.py "<section heading prefix>" "<next heading prefix>" out.inc"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
start, stop, out = sys.argv[1], sys.argv[2], sys.argv[3]
sec = doc[doc.index(start):doc.index(stop)]
blocks = re.findall(r"```cpp\n(.*?)```", sec, re.S)
which = int(sys.argv[4]) if len(sys.argv) > 4 else 0
assert len(blocks) > which, (start, len(blocks))
open(out, "w").write("// extracted from INTEGRATION.md (%s) by tools/extract_block.py\n" % start.strip() + blocks[which])

"""Writes the Option 0 binding of INTEGRATION.md (the replacement of the reference's eight FastLIO entry points over lio_hip.h) to a file,
so that oracle/ref_hdl_fastlio.cpp compiles and LINKS exactly the text the document shows."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
sec = doc[doc.index("## 2. Option 0"):doc.index("## 2a.")]
blocks = re.findall(r"```cpp\n(.*?)```", sec, re.S)
assert len(blocks) == 1
open(sys.argv[1], "w").write("// extracted from INTEGRATION.md section 2 (Option 0) by tools/extract_option0.py\n" + blocks[0])

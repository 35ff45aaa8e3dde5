"""Generates tests/golden/kytea_fullwidth_pairs.txt from the reference's KyteaFullwidthFilter match arms
(vaporetto_rules/src/string_filters/kytea_fullwidth.rs:17-113).  Run in the build container only (it reads
/root/reference); the fixture it writes is what the tests use."""
import os
import re

SRC = "/root/reference/vaporetto_rules/src/string_filters/kytea_fullwidth.rs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kytea_fullwidth_pairs.txt")

pairs = re.findall(r"'(\\?.)' => '(.)',", open(SRC, encoding="utf-8").read())
with open(OUT, "w", encoding="utf-8") as f:
    for a, b in pairs:
        a = a[-1] if a.startswith("\\") else a
        f.write("%04X %04X\n" % (ord(a), ord(b)))
print(len(pairs), "pairs")

"""stdin: `llvm-readelf --notes` of a gfx950 code object -> one line per kernel: registers, spills, scratch, static LDS."""
import re
import sys

txt = sys.stdin.read()
for blk in txt.split("- .agpr_count:")[1:]:
    def g(k):
        m = re.search(r"\." + k + r":\s*(\S+)", blk)
        return m.group(1) if m else "?"
    print("%-110s vgpr %4s agpr %3s spill v%s s%s scratch %4s lds %s" % (g("name")[:110], g("vgpr_count"), blk.split()[0], g("vgpr_spill_count"),
                                                                       g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))

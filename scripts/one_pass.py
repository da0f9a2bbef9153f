"""Dev tool: a few rematch passes at C3 (target for rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from daliti_amd import Engine, synth
c = synth.CONFIGS[os.environ.get("CFG", "C3")]
m = synth.make_map(c["M"], c["L"]); s = synth.make_scan(c["beams"], c["az"], c["L"])
_, xp, P = synth.filter_inputs()
e = Engine(cell_size=float(os.environ.get("CELL", "0.5")))
e.map_build(m); e.scan_set(s)
for _ in range(int(os.environ.get("PASSES", "5"))):
    e.residual_pass(xp, True)
e.close()

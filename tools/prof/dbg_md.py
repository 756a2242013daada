import sys, numpy as np
sys.path.insert(0, '/root/repo')
from elprep_amd.engine import Engine
from tests.common import dataset
cfg, b, h, refs, sites = dataset("tiny", 300, 0, 0.0)
e = Engine(h); e.stage(b)
print("staged", e.n, flush=True)
f = e.mark_duplicates(True)
print("marked", int(((f & 0x400) != 0).sum()), flush=True)

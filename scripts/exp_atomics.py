"""Timing experiment: preprocess_forward / scatter with their global atomics switched off (MOSS_EXPERIMENT bit 1 / 2; results are
garbage, memory accesses stay in range).  Prints the kernel-attached stage times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moss_amd import scenes, _lib
from tests import helpers as hp
dev = torch.device("cuda:0")
d = hp.inputs_of(scenes.config3(), "scale_rot")
for _ in range(5): hp.hip_forward(d, dev)
torch.cuda.synchronize()
_lib.profile_enable(None)
for _ in range(30): hp.hip_forward(d, dev)
torch.cuda.synchronize()
p = _lib.profile_read()
print(os.environ.get("MOSS_EXPERIMENT", "0"), {k: round(1e3 * v[0] / max(v[1], 1), 1) for k, v in p.items()})

"""Per-stage HIP-event times of the fused link at bench size, optionally with another build of the
library (used by tools/ablate_link.sh):

    python tools/link_stage_times.py [path/to/libwfhip.so | '' [nsym [fuse]]]
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from waveforms_amd import _hip
if len(sys.argv) > 1 and sys.argv[1]:
    _hip._LIB_PATH = Path(sys.argv[1]).resolve()
from waveforms_amd.link import SOQPSKLink
nsym = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
fuse = int(sys.argv[3]) if len(sys.argv) > 3 else 7
link = SOQPSKLink(nsym, 8, fuse=fuse)
for k in range(3): link.run_block(10.0, seed=1, stream_id=0, event_slot=0)
acc = {}
for k in range(10):
    link.run_block(10.0, seed=1, stream_id=k, event_slot=k % 8)
torch.cuda.synchronize()
for k in range(8):
    for name, ms in link.stage_ms(k).items(): acc[name] = acc.get(name, 0) + ms / 8
print({k: round(v, 4) for k, v in acc.items()}, link.result())

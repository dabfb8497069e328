"""Soak check: error counts of the link agree across fusion levels on full-size bursts, and a stream
gives the same totals for different chunk sizes.

    python tools/soak_check.py
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from waveforms_amd.link import SOQPSKLink, SOQPSKStream  # noqa: E402

bad = 0
for nsym, ebn0, seed, det in [(10_000_000, 0.0, 3, "PT"), (10_000_000, 6.0, 4, "PT"), (7_654_321, 10.0, 5, "PT"), (5_000_003, 8.0, 6, "PAM"), (12_600_000, 12.0, 7, "PT")]:
    res = {}
    for fuse in (0, 3, 7):
        l = SOQPSKLink(nsym, 8, fuse=fuse, detector=det)
        l.run_block(ebn0, seed=seed, stream_id=seed * 11, skip_bits=12345 * seed)
        res[fuse] = l.result(); del l
    ok = res[0] == res[3] == res[7]
    bad += not ok
    print(nsym, ebn0, det, res[7], "OK" if ok else f"MISMATCH {res}")
for chunk in (1 << 20, 3 << 19, 1 << 22):
    st = SOQPSKStream(30_000_000, chunk, 8)
    print("stream", chunk, st.run(9.0, seed=2, stream_id=5))
print("bad", bad)

"""Soak of the cascading repairs at scale: the same trial blocks through every link with a chunk warm-up of TWO rows (most chunks
miss it; at low Eb/N0 a repair moves the chunk's end and the next chunk follows), with the library's default and with a long one
(nothing to repair): error
counts must be identical, nothing unproven.  repair_soak.py does the same at the operating points' own warm-ups.
    python tools/cascade_soak.py [--blocks 20]"""
import argparse, json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

ap = argparse.ArgumentParser()
ap.add_argument("--blocks", type=int, default=20)
a = ap.parse_args()
import torch
from waveforms_amd import device as dev
from waveforms_amd.link import CPMLink, SOQPSKLink
from waveforms_amd.viterbi import cpm

nsym = 10_000_000
cases = [("soqpsk PT", lambda w: SOQPSKLink(nsym, 8, warmup=w, fuse=47, private_ctx=True), 1, (-8.0, 0.0, 6.0)),
         ("soqpsk PAM", lambda w: SOQPSKLink(nsym, 8, warmup=w, fuse=47, private_ctx=True, detector="PAM"), 1, (-8.0, 2.0)),
         ("pcmfm", lambda w: CPMLink(nsym, 8, waveform="pcmfm", warmup=w, fuse=42, private_ctx=True), 1, (-8.0, 0.0, 2.0)),
         ("multih 16", lambda w: CPMLink(nsym, 8, waveform="multih", warmup=w, fuse=42, private_ctx=True), 2, (-8.0, 0.0, 4.0)),
         # (round 6: the same two links with the matched filters inside the detector — fuse bit 7 — whose repairs rebuild their rows from the samples)
         ("multih 16 samples", lambda w: CPMLink(nsym, 8, waveform="multih", warmup=w, fuse=42 | 128, private_ctx=True), 2, (-8.0, 0.0, 4.0, 10.0)),
         ("multih 256 samples", lambda w: CPMLink(nsym, 8, waveform="multih", spec=cpm.ARTM_256, warmup=w, fuse=10 | 128, private_ctx=True), 2, (0.0,)),
         ("multih 64", lambda w: CPMLink(nsym, 8, waveform="multih", spec=cpm.ARTM_64, warmup=w, fuse=42, private_ctx=True), 2, (0.0,)),
         ("multih 256", lambda w: CPMLink(nsym, 8, waveform="multih", spec=cpm.ARTM_256, warmup=w, fuse=10, private_ctx=True), 2, (0.0,))]
ok = True
for name, make, bps, ebn0s in cases:
    blocks = max(2, a.blocks // 10) if "256" in name else a.blocks
    for ebn0 in ebn0s:
        res = {}
        for w in (2, 0, 640):                      # (0: the library's default warm-up)
            link = make(w)
            dev.viterbi_repaired(reset=True, ctx=link._ctx); dev.viterbi_cascaded(reset=True, ctx=link._ctx)
            for b in range(blocks):
                link.run_block(ebn0, seed=13, stream_id=b, skip_bits=(b % 64) * nsym * bps)
            r = link.result()                      # raises if a chunk was left unproven
            res[w] = (r, dev.viterbi_repaired(reset=True, ctx=link._ctx), dev.viterbi_cascaded(reset=True, ctx=link._ctx))
            del link
        row = {"link": name, "ebn0_db": ebn0, "blocks": blocks, "counts_warmup_2": res[2][0], "repaired": res[2][1], "handed_on": res[2][2],
               "counts_default_warmup": res[0][0], "repaired_default": res[0][1],
               "counts_warmup_640": res[640][0], "repaired_640": res[640][1], "identical": res[2][0] == res[640][0] == res[0][0]}
        ok = ok and row["identical"]
        print(json.dumps(row), flush=True)
print("ALL IDENTICAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)

"""Soak of the CPM detector's repair launch: the same trial blocks through the link with a SHORT chunk warm-up (thousands of
chunks repaired on the device) and with a long one (none): error counts must be identical, no chunk left unproven.
    python tools/repair_soak.py [--blocks 100]"""
import argparse, json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

ap = argparse.ArgumentParser()
ap.add_argument("--blocks", type=int, default=100)
a = ap.parse_args()
import torch
from waveforms_amd import device as dev
from waveforms_amd.link import CPMLink

nsym = 10_000_000
out = []
for waveform, short, ebn0s in (("pcmfm", 64, (2.0, 6.0, 10.0)), ("multih", 48, (8.0, 10.0)), ("multih", 96, (0.0, 4.0))):
    for ebn0 in ebn0s:
        res = {}
        for w in (short, 384):
            link = CPMLink(nsym, 8, waveform=waveform, warmup=w, fuse=42, private_ctx=True)
            dev.viterbi_repaired(reset=True, ctx=link._ctx)
            for b in range(a.blocks):
                link.run_block(ebn0, seed=11, stream_id=b, skip_bits=(b % 64) * nsym * link.spec.bits_per_symbol)
            r = link.result()                      # raises if a chunk was left unproven
            res[w] = (r, dev.viterbi_repaired(reset=True, ctx=link._ctx))
            del link
        row = {"waveform": waveform, "ebn0_db": ebn0, "blocks": a.blocks, "short_warmup": short, "counts_short": res[short][0],
               "repaired_short": res[short][1], "counts_long": res[384][0], "repaired_long": res[384][1],
               "identical": res[short][0] == res[384][0]}
        out.append(row)
        print(json.dumps(row), flush=True)
        assert row["identical"], row
Path("gpurun_out").mkdir(exist_ok=True)
Path("gpurun_out/r03_repair_soak.json").write_text(json.dumps(out, indent=1) + "\n")

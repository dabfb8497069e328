"""Chunk warm-up of the generic CPM detector: chunks whose start state was not bitwise the sequential
detector's (the launch's own proof) and were REPAIRED by the call's second launch, and chunks left unproven, per
warm-up length and Eb/N0; detector time (both launches) at each point.  WF_CPM_NO_REPAIR=1: the proof alone.
    python tools/cpm_warmup_scan.py [--waveform multih] [--symbols-per-point 4e7]"""
import argparse, json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--waveform", default="multih")
    ap.add_argument("--symbols-per-point", type=float, default=6.4e8, help="1e6 chunks of 640 calls per point by default")
    ap.add_argument("--warmups", default="32,48,64,96,128,192,256,320,384,512")
    ap.add_argument("--ebn0", default="0,2,4,6,8,10,12")
    ap.add_argument("--tag", default="r03_cpm_warmup_scan")
    a = ap.parse_args()
    import torch
    from waveforms_amd import device as dev
    from waveforms_amd.link import CPMLink
    nsym = 10_000_000
    blocks = max(1, int(a.symbols_per_point / nsym))
    out = []
    for w in [int(x) for x in a.warmups.split(",")]:
        link = CPMLink(nsym, 8, waveform=a.waveform, warmup=w)
        row = {"waveform": a.waveform, "warmup": w, "blocks_per_point": blocks, "symbols_per_block": nsym, "unmerged_by_ebn0": {}, "repaired_by_ebn0": {},
               "viterbi_ms_by_ebn0": {}}
        for e in [int(x) for x in a.ebn0.split(",")]:
            dev.viterbi_unmerged(reset=True, ctx=link._ctx)
            dev.viterbi_repaired(reset=True, ctx=link._ctx)
            for b in range(blocks):
                link.run_block(float(e), seed=5, stream_id=(e << 20) | b, skip_bits=b * nsym * link.spec.bits_per_symbol)
            torch.cuda.synchronize()
            row["unmerged_by_ebn0"][e] = dev.viterbi_unmerged(reset=True, ctx=link._ctx)
            row["repaired_by_ebn0"][e] = dev.viterbi_repaired(reset=True, ctx=link._ctx)
            ms = []
            for b in range(4):
                link.run_block(float(e), seed=5, stream_id=(e << 20) | b, skip_bits=b * nsym * link.spec.bits_per_symbol, event_slot=b)
            for b in range(4):
                ms.append(link.stage_ms(b)["viterbi"])
            row["viterbi_ms_by_ebn0"][e] = round(sum(ms[1:]) / 3, 4)
        out.append(row)
        print(json.dumps(row), flush=True)
        del link
    Path("gpurun_out").mkdir(exist_ok=True)
    Path(f"gpurun_out/{a.tag}_{a.waveform}.json").write_text(json.dumps(out, indent=1))

if __name__ == "__main__":
    main()

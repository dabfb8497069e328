"""BASELINE config 5: a continuous SOQPSK-TG stream processed in chunks (wf_link_stream_chunk).

    python tools/stream_bench.py [--total 1e9] [--chunk 4194304] [--ebn0 10] [--detector PT]

HBM use is that of ONE chunk whatever the stream length; prints throughput and the BER.
"""
import argparse
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--total", type=float, default=1e9)
    ap.add_argument("--chunk", type=int, default=1 << 22)
    ap.add_argument("--ebn0", type=float, default=10.0)
    ap.add_argument("--detector", default="PT")
    ap.add_argument("--waveform", default="soqpsk", choices=["soqpsk", "multih", "pcmfm"],
                    help="multih / pcmfm: the CPM link's stream (CPMStream, eager launches on one stream)")
    ap.add_argument("--pn-degree", type=int, default=31)
    ap.add_argument("--graph", action="store_true", help="replay the steady-state chunk as a hipGraph")
    ap.add_argument("--pipelined", action="store_true", help="consecutive chunks on two streams (detector of chunk c under the front end of c + 1)")
    ap.add_argument("--graph-pipelined", type=int, default=0, metavar="K", help="the two-stream pipeline with K interior chunks captured as one hipGraph")
    ap.add_argument("--vit-warmup", type=int, default=-1, help="detector chunk warm-up rows (-1: by Eb/N0, waveforms_amd.link.operating_point_warmup; 0: library default)")
    a = ap.parse_args()
    import torch

    from waveforms_amd.link import SOQPSKStream

    if a.waveform != "soqpsk":
        from waveforms_amd.link import CPMStream, operating_point_warmup as opw

        wu = a.vit_warmup if a.vit_warmup >= 0 else opw(a.waveform, a.ebn0)
        # (ARTM: fuse bit 7 — the chunk's samples instead of its rows, the matched filters inside the detector; ignored for PCM/FM)
        st = CPMStream(int(a.total), a.chunk, 8, waveform=a.waveform, pn_degree=a.pn_degree, warmup=wu, fuse=10 | 128)
        st.run_chunk(0, a.ebn0)
        torch.cuda.synchronize()
        go = st.run_pipelined if a.pipelined else st.run
        for rep in range(2):
            t0 = time.perf_counter()
            se, be, m = go(a.ebn0)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(json.dumps({"workload": f"{a.waveform} continuous stream, {int(a.total):.3g} symbols @8 sps in {st.nchunks} chunks of {a.chunk} "
                                      f"(PN{a.pn_degree}, generic CPM trellis detector)", "mode": "two-stream chunk pipeline" if a.pipelined else "eager launches",
                          "Msym_per_s": round(m / dt / 1e6, 1),
                          "seconds": round(dt, 4), "workspace_GB": round(st.workspace_bytes / 1e9, 3), "symbols": m, "symbol_errors": se,
                          "bit_errors": be, "ebn0_db": a.ebn0}))
        return

    from waveforms_amd.link import operating_point_warmup, soqpsk_warmup_param
    wu = soqpsk_warmup_param(a.vit_warmup if a.vit_warmup >= 0 else operating_point_warmup("soqpsk", a.ebn0))    # result() raises if a chunk is unproven
    st = SOQPSKStream(int(a.total), a.chunk, 8, detector=a.detector, pn_degree=a.pn_degree, warmup=wu)
    st.run_chunk(0, a.ebn0)           # warm-up (allocations, code objects)
    torch.cuda.synchronize()
    go = st.run_graph if a.graph else (st.run_pipelined if a.pipelined else st.run)
    if a.graph_pipelined:
        go = lambda *args: st.run_graph_pipelined(*args, chunks_per_graph=a.graph_pipelined)   # noqa: E731
    for rep in range(2):              # the first pass also pays the mode's own allocations (second context, graph capture)
        t0 = time.perf_counter()
        se, be, m = go(a.ebn0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(json.dumps({"workload": f"SOQPSK-TG continuous stream, {int(a.total):.3g} symbols @8 sps in {st.nchunks} chunks "
                                  f"of {a.chunk} (PN{a.pn_degree}, {a.detector} detector)",
                      "mode": f"two-stream chunk pipeline captured as hipGraphs of {a.graph_pipelined} chunks" if a.graph_pipelined else "hipGraph replay" if a.graph else ("two-stream chunk pipeline" if a.pipelined else "eager launches"), "Msym_per_s": round(m / dt / 1e6, 1), "seconds": round(dt, 4),
                      "workspace_GB": round(st.workspace_bytes / 1e9, 3), "symbols": m, "bit_errors": be,
                      "ber": be / max(m, 1), "ebn0_db": a.ebn0}))


if __name__ == "__main__":
    main()

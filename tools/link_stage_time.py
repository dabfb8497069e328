"""Per-stage HIP-event times of the device link WITHOUT the result checks (timing aid for the ablation
builds of tools/ab_kernel.sh, whose outputs are wrong by construction).

    python tools/link_stage_time.py [--waveform soqpsk|multih|pcmfm] [--nsym 10000000] [--steps 20]"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--waveform", default="soqpsk")
    ap.add_argument("--nsym", type=int, default=10_000_000)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--fuse", type=int, default=-1)
    ap.add_argument("--label", default="")
    ap.add_argument("--detector", default="PT", choices=["PT", "PAM"])
    ap.add_argument("--sps", type=int, default=8)
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="wf_ctx options, e.g. cpm_chunk_calls=320")
    a = ap.parse_args()
    import torch

    from waveforms_amd import _hip

    _hip.apply_option_args(a.opt)

    from waveforms_amd.link import CPMLink, SOQPSKLink, operating_point_warmup, soqpsk_warmup_param

    if a.waveform == "soqpsk":
        link = SOQPSKLink(a.nsym, a.sps, pn_degree=23, warmup=soqpsk_warmup_param(operating_point_warmup("soqpsk", 10.0)), fuse=15 if a.fuse < 0 else a.fuse,
                          detector=a.detector)
    else:
        link = CPMLink(a.nsym, a.sps, waveform=a.waveform, warmup=operating_point_warmup(a.waveform, 10.0), fuse=10 if a.fuse < 0 else a.fuse)
    acc = {}
    for k in range(a.steps + 3):
        link.run_block(10.0, seed=1, stream_id=k, event_slot=0)
        torch.cuda.synchronize()
        if k >= 3:
            for name, ms in link.stage_ms(0).items():
                acc[name] = acc.get(name, 0.0) + ms / a.steps
    print(f"[{a.label:44s}] " + " ".join(f"{k}={v:.4f}" for k, v in acc.items() if v > 0) + f"  sum={sum(acc.values()):.4f}", flush=True)


if __name__ == "__main__":
    main()

"""Time wf_cpm_viterbi_detect alone on random rows (tuning / ablation aid).
    python tools/cpm_vit_time.py [--n 10000000] [--spec artm|pcmfm] [--warmup W] [--reps 5] [--opt cpm_form=1] [--opt cpm_chunk_calls=320]
--opt KEY=VALUE: wf_ctx options (include/wfhip.h wf_option): cpm_form 1 = row form, 2 = lane form; cpm_chunk_calls."""
import argparse, ctypes, json, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--spec", default="artm", choices=["artm", "pcmfm"])
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE")
    a = ap.parse_args()
    import torch
    from waveforms_amd import _hip, device as dev
    _hip.apply_option_args(a.opt)
    from waveforms_amd.viterbi import cpm
    spec = cpm.ARTM_16 if a.spec == "artm" else cpm.PCMFM_10
    rows = torch.randn((a.n, spec.nfilt, 2), dtype=torch.float64, device="cuda")
    rows[:, 0, 0] += 2.0
    out = _hip.zeros(a.n + 16, "uint8")
    cfg, rot = spec.c_config(), _hip.to_device(cpm.rotation_table(spec))
    def run():
        _hip.check(_hip.lib().wf_cpm_viterbi_detect(_hip.ctx(), ctypes.byref(cfg), _hip.ptr(rot), _hip.ptr(rows), a.n, a.warmup, _hip.ptr(out), None, _hip.stream()))
    run(); torch.cuda.synchronize()
    dev.viterbi_unmerged(reset=True); dev.viterbi_repaired(reset=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps): run()
    e1.record(); torch.cuda.synchronize()
    print(json.dumps({"spec": a.spec, "n": a.n, "warmup": a.warmup, "opt": a.opt, "kernel": cpm.detector_kernel_name(spec, a.n, a.warmup),
                      "ms_per_call": round(e0.elapsed_time(e1) / a.reps, 4),
                      "chunks_unproven": int(dev.viterbi_unmerged(reset=True)), "chunk_repairs": int(dev.viterbi_repaired(reset=True)),
                      "chunk_repairs_handed_on": int(dev.viterbi_cascaded(reset=True)),
                      "checksum": int(out[:a.n].to(torch.int64).sum().item())}))

if __name__ == "__main__":
    main()

"""Time wf_cpm_viterbi_detect alone on random rows (tuning / ablation aid).
    python tools/cpm_vit_time.py [--n 10000000]"""
import argparse, ctypes, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=10_000_000); a = ap.parse_args()
    import torch
    from waveforms_amd import _hip
    from waveforms_amd.viterbi import cpm
    spec = cpm.ARTM_16
    rows = torch.randn((a.n, spec.nfilt, 2), dtype=torch.float64, device="cuda")
    rows[:, 0, 0] += 2.0
    out = _hip.zeros(a.n + 16, "uint8")
    cfg, rot = spec.c_config(), _hip.to_device(cpm.rotation_table(spec))
    def run():
        _hip.check(_hip.lib().wf_cpm_viterbi_detect(_hip.ctx(), ctypes.byref(cfg), _hip.ptr(rot), _hip.ptr(rows), a.n, 0, _hip.ptr(out), None, _hip.stream()))
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    print("cpm_viterbi ms per launch:", round(e0.elapsed_time(e1) / 5, 4))

if __name__ == "__main__":
    main()

"""Same-box A/B of bench.py over fuse values: driver-timed ms, steady-state ms, bit errors of the steady loop.
    python tools/ab_fuse.py 47 111 [-- extra bench flags per workload ...]"""
import json
import subprocess
import sys

fuses = [a for a in sys.argv[1:] if a.isdigit()]
for wl in ([], ["--detector", "PAM"], ["--sps", "10"], ["--waveform", "multih"], ["--waveform", "pcmfm"]):
    for rep in range(2):
        for f in fuses:
            out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--overlap-streams", "0", "--fuse", f] + wl,
                                 capture_output=True, text=True)
            if out.returncode:
                print(wl, f, "FAILED", out.stderr[-400:], flush=True)
                continue
            d = json.loads(out.stdout.strip().splitlines()[-1])
            s = d["steady_state"]
            print(wl, f, d["ms_per_step"], d["value"], "steady", s["ms_per_step"], s["value"], d["ber"].get("bit_errors"), s["bit_errors"],
                  s["detector_chunks_unproven"], flush=True)

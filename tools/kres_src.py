"""Register / spill figures of selected kernels of ONE source file as the compiler reports them (no link, no library):
    python tools/kres_src.py wf_modulate.hip 'mod_chan_bank_kernel<9, 0, 8>' ... [-D NAME=VALUE ...]
A quick look while editing a kernel; the shipped figures are tools/kernel_resources.py's (from the library's notes)."""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    from waveforms_amd.csrc.build import FLAGS

    src, want, defs = sys.argv[1], [], []
    it = iter(sys.argv[2:])
    for a in it:
        if a == "-D":
            defs.append("-D" + next(it))
        else:
            want.append(a)
    cmd = ["/opt/rocm/bin/hipcc", *[f for f in FLAGS if f != "-fPIC"], *defs, "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
           "-c", str(ROOT / "waveforms_amd" / "csrc" / src), "-o", "/dev/null"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    blocks = re.split(r"remark: [^\n]*Function Name: ", err)[1:]
    names = subprocess.run(["c++filt"], input="\n".join(b.split()[0] for b in blocks), capture_output=True, text=True).stdout.splitlines()
    for b, dem in zip(blocks, names):
        short = re.sub(r"^void ", "", dem).split("(")[0]
        if want and not any(w in short for w in want):
            continue
        f = {k: int(v) for k, v in re.findall(r"(VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", b)}
        print(f"{short:44s} vgpr {f.get('VGPRs'):4d} sgpr {f.get('SGPRs'):4d} vspill {f.get('VGPRs Spill'):3d} sspill {f.get('SGPRs Spill'):3d} scratch {f.get('ScratchSize [bytes/lane]'):4d}")


if __name__ == "__main__":
    main()

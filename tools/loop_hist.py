"""Instruction histogram of the loops of one kernel in the shipped library (disassembly of the gfx950 code object):
    python tools/loop_hist.py 'mod_chan_bank_kernelILi9ELi0ELi8' [--loop N] [--dump]
Loops are listed by size (bytes); --loop N prints the mnemonic histogram of the N-th largest, --dump its text."""
import argparse, collections, re, subprocess, sys, tempfile
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
import kernel_resources as kr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mangled_substring")
    ap.add_argument("--loop", type=int, default=0)
    ap.add_argument("--dump", action="store_true")
    a = ap.parse_args()
    so = Path(kr.DEFAULT_SO)
    for elf in kr.code_objects(so):
        data = elf if isinstance(elf, (bytes, bytearray)) else Path(elf).read_bytes()
        if a.mangled_substring.encode() not in data:
            continue
        f = tempfile.NamedTemporaryFile(suffix=".o", delete=False)
        f.write(data)
        f.close()
        out = subprocess.run([str(Path(str(kr.LLVM)) / "llvm-objdump"), "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
        lines = out.splitlines()
        start = next(i for i, l in enumerate(lines) if a.mangled_substring in l and l.rstrip().endswith(">:"))
        end = start + 1
        while end < len(lines) and not (lines[end].rstrip().endswith(">:") and "_Z" in lines[end]):
            end += 1
        ins = []
        for l in lines[start + 1:end]:
            m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", l)
            if m:
                ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
        loops = []
        for adr, mn, ops in ins:
            if mn.startswith("s_cbranch") or mn == "s_branch":
                m = re.search(r"(-?\d+)$", ops)
                if m:
                    simm = int(m.group(1))
                    simm -= 65536 if simm > 32767 else 0
                    tgt = adr + 4 + 4 * simm
                    if tgt < adr:
                        loops.append((adr - tgt, tgt, adr))
        loops = sorted(set(loops), reverse=True)
        print(len(ins), "instructions; loops (bytes, start, end):", [(s, hex(x), hex(y)) for s, x, y in loops[:8]])
        size, t0, t1 = loops[a.loop]
        seg = [(adr, mn, ops) for adr, mn, ops in ins if t0 <= adr <= t1]
        c = collections.Counter(mn for _, mn, _ in seg)
        print(f"loop {a.loop}: {hex(t0)}..{hex(t1)}: {len(seg)} instructions")
        for mn, n in c.most_common():
            print(f"  {n:4d} {mn}")
        if a.dump:
            for adr, mn, ops in seg:
                print(f"{adr:x}: {mn} {ops}")
        return
    raise SystemExit("kernel not found")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Resource usage of every kernel in the SHIPPED libwfhip.so, read from the code-object notes.

    python tools/kernel_resources.py [--so PATH] [--json] [--filter SUBSTR] [--loop-spills]

The .so carries one clang offload bundle per translation unit in its .hip_fatbin section; each
bundle holds a gfx950 ELF whose NT_AMDGPU_METADATA note lists, per kernel, .vgpr_count,
.agpr_count, .sgpr_count, .vgpr_spill_count, .sgpr_spill_count, .private_segment_fixed_size
(scratch bytes), .group_segment_fixed_size (static LDS) and .max_flat_workgroup_size.
`tests/test_kernel_resources.py` pins ceilings on these for the hot kernels, so that DESIGN.md
cannot drift from the binary (round-2 verdict, weak #4).

--loop-spills additionally disassembles the code objects and counts, per kernel, the
`v_readlane_b32 s*, v*` / `v_writelane_b32` instructions (SGPR spill traffic) and scratch_load /
scratch_store instructions (VGPR spill traffic) that sit inside a loop (between a label and a
backward branch to it).
"""
from __future__ import annotations

import argparse
import json
import re
import struct
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
DEFAULT_SO = ROOT / "waveforms_amd" / "csrc" / "libwfhip.so"
LLVM = Path("/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"

FIELDS = (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count",
          ".private_segment_fixed_size", ".group_segment_fixed_size", ".max_flat_workgroup_size",
          ".wavefront_size")


def _section(so: Path, name: str) -> bytes:
    """Raw bytes of an ELF section of the host library (no objcopy dependency)."""
    d = so.read_bytes()
    assert d[:4] == b"\x7fELF" and d[4] == 2, "not a 64-bit ELF"
    shoff, = struct.unpack_from("<Q", d, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", d, 0x3A)
    def sh(i):
        return struct.unpack_from("<IIQQQQIIQQ", d, shoff + i * shentsize)
    stro = sh(shstrndx)[4]
    for i in range(shnum):
        n, _t, _f, _a, off, size, *_ = sh(i)
        end = d.index(b"\0", stro + n)
        if d[stro + n:end].decode() == name:
            return d[off:off + size]
    raise KeyError(name)


def code_objects(so: Path) -> list[bytes]:
    """The gfx950 ELF of every bundle in .hip_fatbin."""
    fat = _section(so, ".hip_fatbin")
    out = []
    for m in re.finditer(re.escape(MAGIC), fat):
        base = m.start()
        n, = struct.unpack_from("<Q", fat, base + len(MAGIC))
        p = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", fat, p)
            triple = fat[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "amdgcn" in triple and size:
                out.append(fat[base + off:base + off + size])
    return out


def _demangle(names: list[str]) -> list[str]:
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
        return r.stdout.splitlines()
    except Exception:
        return names


def kernel_table(so: Path = DEFAULT_SO) -> dict[str, dict[str, int]]:
    """{demangled kernel name: {field: value}} for every kernel of the library."""
    table: dict[str, dict[str, int]] = {}
    with tempfile.TemporaryDirectory() as td:
        for i, co in enumerate(code_objects(so)):
            f = Path(td) / f"co{i}.elf"
            f.write_bytes(co)
            txt = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(f)], capture_output=True, text=True, check=True).stdout
            cur: dict[str, int] | None = None
            entries: list[tuple[str, dict[str, int]]] = []
            # the metadata is YAML: kernels are list items ("  - .agpr_count: 0" starts one)
            in_kernels = False
            for ln in txt.splitlines():
                s = ln.strip()
                if s.startswith("amdhsa.kernels:"):
                    in_kernels = True
                    continue
                if in_kernels and s.startswith("amdhsa.") and not s.startswith("amdhsa.kernels"):
                    in_kernels = False
                if not in_kernels:
                    continue
                if re.match(r"^  - \.", ln):          # a new kernel record (two-space list indent)
                    cur = {}
                    entries.append(("", cur))
                    s = s[2:]
                if cur is None:
                    continue
                m = re.match(r"^(\.[a-z_]+):\s*(.*)$", s)
                if not m or not ln.startswith("    ." if not ln.startswith("  - ") else "  - "):
                    continue
                k, v = m.group(1), m.group(2).strip()
                if k == ".name" and not entries[-1][0]:
                    entries[-1] = (v.strip("'\""), cur)
                elif k in FIELDS:
                    try:
                        cur[k.lstrip(".")] = int(v, 0)
                    except ValueError:
                        pass
            names = _demangle([n for n, _ in entries])
            for (raw, rec), dn in zip(entries, names):
                rec["mangled"] = raw  # type: ignore[assignment]
                dn = re.sub(r"^void ", "", dn)
                dn = re.sub(r"\(.*$", "", dn)
                table[dn] = rec
    return table


def loop_spill_counts(so: Path = DEFAULT_SO, name_filter: str = "") -> dict[str, dict[str, int]]:
    """Per kernel: spill-looking instructions in total and inside loops (label .. backward branch)."""
    res: dict[str, dict[str, int]] = {}
    with tempfile.TemporaryDirectory() as td:
        for i, co in enumerate(code_objects(so)):
            f = Path(td) / f"co{i}.elf"
            f.write_bytes(co)
            asm = subprocess.run([str(LLVM / "llvm-objdump"), "-d", "--no-show-raw-insn", str(f)],
                                 capture_output=True, text=True, check=True).stdout
            cur, lines = None, []
            blocks: list[tuple[str, list[str]]] = []
            for ln in asm.splitlines():
                m = re.match(r"^[0-9a-f]+ <([^>]+)>:$", ln)
                if m:
                    lab = m.group(1)
                    if not lab.startswith("L") or cur is None:
                        cur = lab
                        lines = []
                        blocks.append((cur, lines))
                        continue
                lines.append(ln) if cur is not None else None
            names = _demangle([b[0] for b in blocks])
            for (raw, body), dn in zip(blocks, names):
                dn = re.sub(r"\(.*$", "", re.sub(r"^void ", "", dn))
                if name_filter and name_filter not in dn:
                    continue
                # address of every instruction; backward branches mark loops [target, branch]
                insts = []
                for ln in body:
                    m = re.match(r"^\s*(\S.*?)\s*//\s*([0-9A-Fa-f]+):", ln)
                    if m:
                        insts.append((int(m.group(2), 16), m.group(1)))
                # s_branch / s_cbranch_* carry a signed 16-bit dword offset from the next instruction;
                # a backward branch closes a loop [target, branch]
                loops = []
                for addr, txt in insts:
                    m = re.match(r"^s_c?branch\S*\s+(\d+)", txt)
                    if m:
                        imm = int(m.group(1))
                        imm = imm - 65536 if imm >= 32768 else imm
                        ta = addr + 4 + 4 * imm
                        if ta <= addr:
                            loops.append((ta, addr))
                def in_loop(a):
                    return any(lo <= a <= hi for lo, hi in loops)
                def in_nested(a):                       # inside a loop that itself sits in a loop (e.g. row loop in tile loop)
                    return sum(lo <= a <= hi for lo, hi in loops) >= 2
                rl = [a for a, t in insts if re.match(r"^v_readlane_b32\s+s", t)]
                wl = [a for a, t in insts if t.startswith("v_writelane_b32")]
                sl = [a for a, t in insts if t.startswith("scratch_load")]
                ss = [a for a, t in insts if t.startswith("scratch_store")]
                res[dn] = {"instructions": len(insts), "loops": len(loops),
                           "v_readlane": len(rl), "v_readlane_in_loop": sum(map(in_loop, rl)),
                           "v_readlane_in_nested_loop": sum(map(in_nested, rl)),
                           "scratch_in_nested_loop": sum(map(in_nested, sl)) + sum(map(in_nested, ss)),
                           "v_writelane": len(wl), "v_writelane_in_loop": sum(map(in_loop, wl)),
                           "scratch_load": len(sl), "scratch_load_in_loop": sum(map(in_loop, sl)),
                           "scratch_store": len(ss), "scratch_store_in_loop": sum(map(in_loop, ss))}
    return res


def waves_per_simd(vgprs: int, agprs: int = 0) -> int:
    """gfx950: 512 unified registers per lane and SIMD, allocated in blocks of 8, at most 8 waves."""
    tot = max(1, vgprs + agprs)
    tot = (tot + 7) // 8 * 8
    return max(1, min(8, 512 // tot))


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--so", default=str(DEFAULT_SO))
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--filter", default="")
    ap.add_argument("--loop-spills", action="store_true")
    a = ap.parse_args()
    tab = kernel_table(Path(a.so))
    if a.filter:
        tab = {k: v for k, v in tab.items() if a.filter in k}
    if a.loop_spills:
        ls = loop_spill_counts(Path(a.so), a.filter)
        for k in tab:
            tab[k].update({("asm_" + kk): vv for kk, vv in ls.get(k, {}).items()})
    if a.json:
        print(json.dumps(tab, indent=1, sort_keys=True))
        return 0
    hdr = f"{'kernel':70s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'lds':>6s} {'waves':>5s}"
    print(hdr)
    for k in sorted(tab):
        r = tab[k]
        print(f"{k[:70]:70s} {r.get('vgpr_count', 0):5d} {r.get('agpr_count', 0):5d} {r.get('sgpr_count', 0):5d} "
              f"{r.get('vgpr_spill_count', 0):6d} {r.get('sgpr_spill_count', 0):6d} {r.get('private_segment_fixed_size', 0):7d} "
              f"{r.get('group_segment_fixed_size', 0):6d} {waves_per_simd(r.get('vgpr_count', 0), r.get('agpr_count', 0)):5d}"
              + (f"   readlane {r.get('asm_v_readlane', 0)} ({r.get('asm_v_readlane_in_loop', 0)} in loops, {r.get('asm_v_readlane_in_nested_loop', 0)} nested), scratch ld/st "
                 f"{r.get('asm_scratch_load', 0)}/{r.get('asm_scratch_store', 0)} ({r.get('asm_scratch_load_in_loop', 0)}/"
                 f"{r.get('asm_scratch_store_in_loop', 0)} in loops, {r.get('asm_scratch_in_nested_loop', 0)} nested)" if a.loop_spills else ""))
    return 0


if __name__ == "__main__":
    sys.exit(main())

"""Instruction mix of one kernel in a hipcc -save-temps .s file, whole function and its hottest loop (the longest basic-block
run between a label and a backward branch to it).   python tools/isa_mix.py file.s <substring of the mangled name> [...]"""
import re
import sys
from collections import Counter

KEYS = ["v_fma_f64", "v_add_f64", "v_mul_f64", "v_min_f64", "v_cmp_lt_f64_e32", "v_cmp_lt_f64_e64", "v_cndmask_b32_e64", "v_cndmask_b32_e32",
        "s_load_dwordx2", "s_load_dwordx4", "s_load_dwordx8", "s_load_dwordx16", "global_load_dwordx4", "global_load_dwordx2",
        "global_load_lds_dwordx4", "scratch_load_dword", "scratch_store_dword", "scratch_load_dwordx2", "scratch_store_dwordx2", "v_readlane_b32", "v_writelane_b32", "ds_read_b128", "ds_read_b64",
        "v_accvgpr_write_b32", "v_accvgpr_read_b32", "s_waitcnt", "v_mov_b32_e32", "s_nop"]


def main():
    src = open(sys.argv[1]).read().split("\n")
    for want in sys.argv[2:]:
        start = next(i for i, l in enumerate(src) if l.startswith("_Z") and want in l.split(":")[0] and l.rstrip().split(";")[0].rstrip().endswith(":"))
        end = next(i for i in range(start, len(src)) if src[i].startswith(".Lfunc_end"))
        body = src[start:end]
        ins = [(i, l.strip()) for i, l in enumerate(body) if l.strip() and not l.strip().startswith((";", ".", "_Z")) or re.match(r"\.LBB\d+_\d+:", l.strip())]
        labels = {l.split(":")[0]: i for i, l in ins if l.startswith(".LBB")}
        best = None
        for i, l in ins:
            m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"s_branch\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                span = (labels[m.group(1)], i)
                if best is None or span[1] - span[0] > best[1] - best[0]:
                    best = span
        for title, lo, hi in (("function", 0, len(body)), ("hottest loop", *(best or (0, 0)))):
            c = Counter(l.split()[0] for i, l in ins if lo <= i <= hi and not l.startswith(".LBB"))
            tot = sum(c.values())
            valu = sum(v for k, v in c.items() if k.startswith("v_"))
            print(f"{want} [{title}]: {tot} instructions, {valu} vector;", ", ".join(f"{k} {c[k]}" for k in KEYS if c[k]))


if __name__ == "__main__":
    main()

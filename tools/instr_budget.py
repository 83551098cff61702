#!/usr/bin/env python3
"""Instruction budget of a stretch of a kernel's gfx950 assembly, by class.

    hipcc ... -S --cuda-device-only pt_kernels.hip -o k.s
    python tools/instr_budget.py k.s SYMBOL name:first_line:last_line [name:first:last ...]

Line numbers count from the kernel's own label (line 1 = the symbol).  Classes: packed / scalar slab arithmetic
(v_pk_add / v_pk_mul / min3 / max3 / the slack fma), comparisons and mask logic, selects (v_cndmask: hit masks and the
ordering network), integer / address arithmetic, vector memory, LDS, scratch, scalar ALU and control (exec masks,
branches), waits and nops.  A stretch with branches is counted as laid out, every block once."""
import re
import sys

CLASSES = [
    ("vector memory (global)", r"^(global_|buffer_|flat_)"),
    ("LDS", r"^ds_"),
    ("scratch", r"^scratch_"),
    ("packed fp32 arithmetic", r"^v_pk_(add|mul|fma)_f32"),
    ("min3 / max3 / min / max", r"^v_(min3|max3|min|max|med3)_f32"),
    ("fp32 arithmetic (other)", r"^v_(fma|mul|add|sub|mac|fmac|rcp|rsq|sqrt|div_scale|div_fmas|div_fixup|mad|fract|floor|trunc|rndne|ldexp)_\w*f32"),
    ("comparisons", r"^v_cmp"),
    ("selects (v_cndmask)", r"^v_cndmask"),
    ("integer / address / moves", r"^v_(add|sub|lshl|lshr|ashr|and|or|xor|not|mov|mad_u|mad_i|mul_lo|mul_u|mul_i|mul_hi|bfe|bfi|perm|lshl_add|add_lshl|lshl_or|bitop|pk_mov|readlane|writelane|readfirstlane|mbcnt|cvt|accvgpr)"),
    ("scalar ALU and exec-mask logic", r"^s_(and|or|xor|andn2|orn2|mov|cmp|cselect|add|sub|mul|lshl|lshr|bcnt|ff1|not|bfe|sext|ashr|min|max|load|movk|addc|subb|getpc|bitcmp|flbit|brev|pack)"),
    ("branches", r"^s_(cbranch|branch|setpc|swappc|endpgm)"),
    ("waits / nops", r"^s_(waitcnt|nop|sleep|barrier|setprio)"),
]


def kernel_lines(path, symbol):
    out, on = [], False
    for line in open(path):
        if line.startswith(symbol):
            on = True
        if on:
            out.append(line.rstrip("\n"))
            if ".end_amdhsa_kernel" in line:
                break
    return out


def classify(lines):
    counts = {name: 0 for name, _ in CLASSES}
    counts["other"] = 0
    total = 0
    for line in lines:
        t = line.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":") or re.match(r"^\.?LBB", t):
            continue
        op = t.split()[0]
        total += 1
        for name, rx in CLASSES:
            if re.match(rx, op):
                counts[name] += 1
                break
        else:
            counts["other"] += 1
    return total, counts


def main():
    path, symbol = sys.argv[1], sys.argv[2]
    lines = kernel_lines(path, symbol)
    print("kernel %s: %d assembly lines" % (symbol[:60], len(lines)))
    for spec in sys.argv[3:]:
        name, a, b = spec.rsplit(":", 2)
        total, counts = classify(lines[int(a) - 1:int(b)])
        valu = sum(v for k, v in counts.items() if k not in ("scalar ALU and exec-mask logic", "branches", "waits / nops", "vector memory (global)", "LDS", "scratch", "other"))
        print("\n%s (lines %s-%s): %d instructions, %d of them vector ALU" % (name, a, b, total, valu))
        for k, v in counts.items():
            if v:
                print("    %-34s %4d" % (k, v))


if __name__ == "__main__":
    main()

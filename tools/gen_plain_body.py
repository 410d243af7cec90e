#!/usr/bin/env python3
"""Generate candidate schedules of the (all-plain) interaction body for K = 2 on the scalar-cache route, as a header for
`tools/build_variants.sh "name:-include <header> -include $PWD/tools/exp_body_hook.h"` (the hook replaces kernels.hip's interact8).

  grouped S    S sources x 2 receivers per statement: all heads, ONE priority window with all 2*S v_rsq_f32, all tails
  twosrc       the shipped paired body twice in one statement (hipcc's pad after every second source)
  tailmix      the shipped paired body with its two tails interleaved instruction by instruction

Measured at N = 2^20 (profiles/r02_ab_plain_body_schedules.txt): grouped 1 = the shipped body; grouped 2: +0.2 %,
grouped 4: +0.5 % over it; twosrc and tailmix: within noise."""
import sys


def head(dx, dy, q, s, k):
    return [f"v_sub_f32 v{dx}, %[sx{s}], %[px{k}]", f"v_sub_f32 v{dy}, %[sy{s}], %[py{k}]",
            f"v_fma_f32 v{q}, v{dx}, v{dx}, %[r{k}]", f"v_fmac_f32 v{q}, v{dy}, v{dy}"]


def tail(dx, dy, q, s, k, u=36, t=32):
    return [f"v_mul_f32 v{u}, %[g{s}], v{q}", f"v_mul_f32 v{t}, v{q}, v{q}", f"v_mul_f32 v{u}, v{u}, v{t}",
            f"v_fmac_f32 %[ax{k}], v{dx}, v{u}", f"v_fmac_f32 %[ay{k}], v{dy}, v{u}"]


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "grouped"
    lines, clob, per_stmt = [], {30, 31, 32, 33, 36, 37}, 1
    if mode == "grouped":
        per_stmt = int(sys.argv[2]) if len(sys.argv) > 2 else 1
        assert per_stmt in (1, 2), "kernels.hip binds two sources per statement at most"
        regs, nxt = [], 38
        for s in range(per_stmt):
            for k in range(2):
                if not regs:
                    regs.append((30, 31, 33))
                else:
                    regs.append((nxt, nxt + 1, nxt + 2))
                    nxt += 3
        it = iter(regs)
        for s in range(per_stmt):
            for k in range(2):
                lines += head(*next(it), s, k)
        lines += ["s_setprio 3"] + [f"v_rsq_f32 v{q}, v{q}" for _, _, q in regs] + ["s_setprio 0"]
        it = iter(regs)
        for s in range(per_stmt):
            for k in range(2):
                lines += tail(*next(it), s, k)
        clob |= {x for t in regs for x in t}
    elif mode == "twosrc":
        per_stmt = 2
        for s in range(2):
            lines += head(30, 31, 33, s, 0) + head(38, 39, 40, s, 1)
            lines += ["s_setprio 3", "v_rsq_f32 v33, v33", "v_rsq_f32 v40, v40", "s_setprio 0"]
            lines += tail(30, 31, 33, s, 0) + tail(38, 39, 40, s, 1)
        clob |= {38, 39, 40}
    elif mode == "tailmix":
        a, b = tail(30, 31, 33, 0, 0, 36, 32), tail(38, 39, 40, 0, 1, 41, 42)
        lines = head(30, 31, 33, 0, 0) + head(38, 39, 40, 0, 1)
        lines += ["s_setprio 3", "v_rsq_f32 v33, v33", "v_rsq_f32 v40, v40", "s_setprio 0"]
        lines += [x for pair in zip(a, b) for x in pair]
        clob |= {38, 39, 40, 41, 42}
    else:
        sys.exit(__doc__)
    print(f"#define NB_EXPGEN_S {per_stmt}")
    print("#define NB_EXPGEN_ASM \\")
    print(" \\\n".join(f'    "{l}\\n\\t"' for l in lines[:-1]) + f' \\\n    "{lines[-1]}"')
    print("#define NB_EXPGEN_CLOBBERS " + ", ".join(f'"v{c}"' for c in sorted(clob)))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Generate tools/ubench3.hip: candidate hand-scheduled inner bodies (8 sources x 4 receivers per wave) for the
step kernel, with fixed physical registers, to find the cheapest legal ordering on gfx950 before writing the real one.

Register map (per lane): receiver k: X,Y = v[2k:2k+1]; R = v[8+k]; acc pair = v[12+2k:13+2k]
slot j (0..7): D pair (dx,dy) = v[20+2j:21+2j]; Q (d2/inv/gi/f) = v[36+2j]; T (inv^2) = v[37+2j]
sources u (0..7): x,y = s[16+2u:17+2u]; G*m = s[32+u]
"""
X = lambda k: f"v{2*k}"
Y = lambda k: f"v{2*k+1}"
XY = lambda k: f"v[{2*k}:{2*k+1}]"
R = lambda k: f"v{8+k}"
AX = lambda k: f"v{12+2*k}"
AY = lambda k: f"v{13+2*k}"
AXY = lambda k: f"v[{12+2*k}:{13+2*k}]"
DX = lambda j: f"v{20+2*j}"
DY = lambda j: f"v{21+2*j}"
DXY = lambda j: f"v[{20+2*j}:{21+2*j}]"
Q = lambda j: f"v{36+2*j}"
T = lambda j: f"v{37+2*j}"
QT = lambda j: f"v[{36+2*j}:{37+2*j}]"
SX = lambda u: f"s{16+2*u}"
SY = lambda u: f"s{17+2*u}"
SXY = lambda u: f"s[{16+2*u}:{17+2*u}]"
SG = lambda u: f"s{32+u}"


def head(u, k, j, packed):
    if packed:
        a = [f"v_pk_add_f32 {DXY(j)}, {SXY(u)}, {XY(k)} neg_lo:[0,1] neg_hi:[0,1]"]
    else:
        a = [f"v_sub_f32 {DX(j)}, {SX(u)}, {X(k)}", f"v_sub_f32 {DY(j)}, {SY(u)}, {Y(k)}"]
    return a + [f"v_fma_f32 {Q(j)}, {DX(j)}, {DX(j)}, {R(k)}", f"v_fmac_f32 {Q(j)}, {DY(j)}, {DY(j)}"]


PRIO = False


def rsq(j):
    if PRIO:
        return ["s_setprio 3", f"v_rsq_f32 {Q(j)}, {Q(j)}", "s_setprio 0"]
    return [f"v_rsq_f32 {Q(j)}, {Q(j)}"]


def rsq_group(js):
    if PRIO:
        return ["s_setprio 3"] + [f"v_rsq_f32 {Q(j)}, {Q(j)}" for j in js] + ["s_setprio 0"]
    return [f"v_rsq_f32 {Q(j)}, {Q(j)}" for j in js]


def tail(u, k, j, packed):
    a = [f"v_mul_f32 {T(j)}, {Q(j)}, {Q(j)}", f"v_mul_f32 {Q(j)}, {SG(u)}, {Q(j)}", f"v_mul_f32 {Q(j)}, {Q(j)}, {T(j)}"]
    if packed:
        a += [f"v_pk_fma_f32 {AXY(k)}, {DXY(j)}, {QT(j)}, {AXY(k)} op_sel_hi:[1,0,1]"]
    else:
        a += [f"v_fmac_f32 {AX(k)}, {DX(j)}, {Q(j)}", f"v_fmac_f32 {AY(k)}, {DY(j)}, {Q(j)}"]
    return a


def interleave(lists):
    """round-robin merge: instruction i of every list before instruction i+1 of any (max ILP distance)"""
    out = []
    for i in range(max(len(l) for l in lists)):
        for l in lists:
            if i < len(l):
                out.append(l[i])
    return out


def body(order, packed):
    ins = []
    if order == "serial":
        for u in range(8):
            for k in range(4):
                ins += head(u, k, k, packed) + rsq(k) + ["s_nop 0"] + tail(u, k, k, packed)  # trans -> VALU use needs 1 wait state
    elif order == "phase4":      # per source: heads of 4 receivers interleaved, 4 rsq, tails interleaved
        for u in range(8):
            ins += interleave([head(u, k, k, packed) for k in range(4)])
            ins += rsq_group(range(4))
            ins += interleave([tail(u, k, k, packed) for k in range(4)])
    elif order == "phase8":      # per 2 sources: 8 heads, 8 rsq, 8 tails
        for u in range(0, 8, 2):
            sl = [(u + d, k, 4 * d + k) for d in range(2) for k in range(4)]
            ins += interleave([head(uu, k, j, packed) for (uu, k, j) in sl])
            ins += rsq_group([j for (_, _, j) in sl])
            ins += interleave([tail(uu, k, j, packed) for (uu, k, j) in sl])
    elif order == "pipe4":       # rotated: head(u+1) | rsq(u+1) | tail(u); slots alternate between two banks of 4
        def H(u): return interleave([head(u, k, 4 * (u & 1) + k, packed) for k in range(4)])
        def S(u): return [x for k in range(4) for x in rsq(4 * (u & 1) + k)]
        def Tl(u): return interleave([tail(u, k, 4 * (u & 1) + k, packed) for k in range(4)])
        ins += H(0) + S(0)
        for u in range(8):
            if u + 1 < 8:
                ins += H(u + 1) + S(u + 1)
            ins += Tl(u)
    elif order == "pipe4b":      # like pipe4 but tail(u) BEFORE head(u+1): rsq(u+1) then tail(u+1) gets latency distance
        def H(u): return interleave([head(u, k, 4 * (u & 1) + k, packed) for k in range(4)])
        def S(u): return [x for k in range(4) for x in rsq(4 * (u & 1) + k)]
        def Tl(u): return interleave([tail(u, k, 4 * (u & 1) + k, packed) for k in range(4)])
        ins += H(0) + S(0)
        for u in range(8):
            if u + 1 < 8:
                ins += H(u + 1)
            ins += Tl(u)
            if u + 1 < 8:
                ins += S(u + 1)
    elif order == "mix":         # what a list scheduler tends to do: one rsq after every head, tails of the previous slot after it
        slots = [(u, k) for u in range(8) for k in range(4)]
        prev = None
        for n, (u, k) in enumerate(slots):
            j = n & 7
            ins += head(u, k, j, packed) + rsq(j)
            if prev is not None:
                ins += tail(*prev, packed)
            prev = (u, k, j)
        ins += tail(*prev, packed)
    return ins


VARIANTS = [(o, True, pr) for o in ("serial", "phase4", "phase8", "mix") for pr in (False, True)]
CLOBBER = ", ".join(f'"v{i}"' for i in range(52)) + ", " + ", ".join(f'"s{i}"' for i in range(16, 40))

src = ['// GENERATED by tools/gen_ubench3.py -- candidate inner bodies of the step kernel (see that file).',
       '#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <cstdlib>',
       '#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\\n", __FILE__, __LINE__, hipGetErrorString(e_)); abort(); } } while (0)',
       '']
for vi, (o, p, pr) in enumerate(VARIANTS):
    PRIO = pr
    ins = body(o, p)
    init = [f"v_cvt_f32_u32 v{i}, v{i}" for i in range(0)]  # nothing
    setup = [f"v_mov_b32 v{i}, {1.0 + 0.37 * i:.3f}" for i in range(12)] + [f"v_mov_b32 v{i}, 0" for i in range(12, 20)]
    setup += [f"s_mov_b32 s{16+i}, {2.0 + 0.11 * i:.3f}" for i in range(24)]
    src.append(f'// {o} {"prio" if pr else "noprio"}: {len(ins)} instructions per 32 interactions')
    src.append(f'__global__ __launch_bounds__(256) void body{vi}(float *out, int iters) {{')
    src.append('    asm volatile("' + '\\n\\t'.join(setup) + f'" ::: {CLOBBER});')
    src.append('    for (int it = 0; it < iters; it++) {')
    src.append('        asm volatile("' + '\\n\\t'.join(ins) + f'" ::: {CLOBBER});')
    src.append('    }')
    src.append('    float s; asm volatile("v_add_f32 %0, v12, v13\\n\\tv_add_f32 %0, %0, v14\\n\\tv_add_f32 %0, %0, v19" : "=v"(s) :: ' + CLOBBER + ');')
    src.append('    if (s == 12345.678f) out[0] = s;')
    src.append('}')
    src.append('')
names = ", ".join(f'"{o} {"prio" if pr else "noprio"}"' for (o, p, pr) in VARIANTS)
fns = ", ".join(f"body{i}" for i in range(len(VARIANTS)))
src.append(f'''int main() {{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    float *out; CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[] = {{{names}}};
    void (*fn[])(float *, int) = {{{fns}}};
    const int iters = 4000, cus = prop.multiProcessorCount;
    for (int v = 0; v < {len(VARIANTS)}; v++)
        for (int wps = 4; wps <= 8; wps *= 2) {{
            dim3 grid(cus * wps), block(256);
            hipLaunchKernelGGL(fn[v], grid, block, 0, 0, out, 50); CK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int r = 0; r < 3; r++) {{
                CK(hipEventRecord(e0, 0)); hipLaunchKernelGGL(fn[v], grid, block, 0, 0, out, iters); CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
            }}
            double inter = (double)iters * 32 * 64 * 4.0 * cus * wps;   // lane-interactions
            double cyc = best * 1e-3 * 2.4e9 / ((double)iters * wps * 32);  // SIMD cycles (at 2.4 GHz) per wave-interaction
            printf("%-16s waves/SIMD %d  %8.3f ms  %.3e interactions/s  %.2f cyc/wave-interaction\\n", names[v], wps, best, inter / (best * 1e-3), cyc);
        }}
    return 0;
}}''')
open(__file__.replace("gen_ubench3.py", "ubench3.hip"), "w").write("\n".join(src) + "\n")
print("wrote ubench3.hip with", len(VARIANTS), "variants")

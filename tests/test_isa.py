"""Static checks on the gfx950 ISA of the step kernels (no GPU needed: hipcc cross-compiles).

The hot loop is hand-written asm (nbody_amd/csrc/kernels.hip, NB_INTERACTION_ASM) and leans on three invariants that
only the GPU parity tests would otherwise notice:
  1. gfx950 needs one wait state between a transcendental (v_rsq_f32) and the VALU instruction that reads its
     result, and hipcc cannot pad inside an asm statement: the `s_setprio 0` that follows the rsq IS that wait state
     (round 1 shipped-then-fixed a variant without it that read stale values);
  2. every step kernel must stay within 64 VGPRs, or a 1024-thread workgroup no longer fits twice on a CU
     (__launch_bounds__(1024, 8));
  3. no scratch: a spill inside the inner loop would sit on the critical path.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    out = tmp_path_factory.mktemp("isa") / "kernels.s"
    src = os.path.join(ROOT, "nbody_amd", "csrc", "kernels.hip")
    # the flags of nbody_amd/csrc/Makefile (HIPFLAGS), device side only, assembly out
    cmd = [HIPCC, "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-Wno-unused-command-line-argument",
           f"-I{ROOT}/include", f"-I{ROOT}/nbody_amd/csrc", "--cuda-device-only", "-S", "-o", str(out), src]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    return out.read_text()


def functions(text):
    """name -> list of instruction lines (labels, directives and comments dropped)."""
    out, name = {}, None
    for line in text.splitlines():
        m = re.match(r"^(_ZN2nb\S+):", line)
        if m:
            name = m.group(1)
            out[name] = []
            continue
        if line.startswith(".Lfunc_end"):
            name = None
            continue
        if name is None:
            continue
        ins = line.split(";")[0].strip()
        if not ins or ins.startswith(".") or ins.endswith(":"):
            continue
        out[name].append(ins)
    return out


def reads_register(ins, reg):
    """Does instruction text `ins` mention VGPR number `reg` (alone or inside a v[a:b] range)?"""
    for m in re.finditer(r"\bv(\d+)\b", ins):
        if int(m.group(1)) == reg:
            return True
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", ins):
        if int(m.group(1)) <= reg <= int(m.group(2)):
            return True
    return False


def test_every_rsq_has_a_wait_state_before_its_consumer(isa):
    fns = {n: body for n, body in functions(isa).items() if "step_kernel" in n}
    assert len(fns) >= 16, sorted(fns)
    total = 0
    for name, body in fns.items():
        for i, ins in enumerate(body):
            if not ins.startswith("v_rsq_f32"):
                continue
            total += 1
            dest = int(re.match(r"v_rsq_f32(?:_e\d+)?\s+v(\d+)", ins).group(1))
            nxt = body[i + 1]
            assert not (nxt.startswith("v_") and reads_register(nxt, dest)), \
                f"{name}: `{ins}` is read by the very next instruction `{nxt}` (no wait state)"
            # the shipped bodies: one rsq (K = 1) or two back to back (K = 2) inside a raised-priority window whose
            # closing s_setprio 0 is the wait state before the first dependent multiply
            assert any(x.startswith("s_setprio 0") for x in body[i + 1:i + 3]), f"{name}: no s_setprio 0 after `{ins}`: {body[i + 1:i + 3]}"
            assert any(x.startswith("s_setprio 3") for x in body[i - 2:i]), f"{name}: rsq not issued at raised priority: {body[i - 2:i]}"
            reader = next(j for j in range(i + 1, len(body)) if body[j].startswith("v_") and reads_register(body[j], dest))
            assert any(x.startswith("s_") for x in body[i + 1:reader]), f"{name}: no scalar slot between `{ins}` and `{body[reader]}`"
    assert total >= 16 * 8, total


def test_step_kernels_fit_the_occupancy_the_launch_bounds_promise(isa):
    meta = re.findall(r"\.name:\s+(\S+)\n\s+\.private_segment_fixed_size:\s+(\d+)\n\s+\.sgpr_count:\s+(\d+)"
                      r"(?:\n.*?)*?\n\s+\.vgpr_count:\s+(\d+)", isa)
    step = [(n, int(scratch), int(sgpr), int(vgpr)) for n, scratch, sgpr, vgpr in meta if "step_kernel" in n]
    assert len(step) >= 16
    for name, scratch, sgpr, vgpr in step:
        assert vgpr <= 64, f"{name}: {vgpr} VGPRs (> 64 halves the occupancy of a 1024-thread workgroup)"
        assert scratch == 0, f"{name}: {scratch} bytes of scratch (spills)"
        assert sgpr <= 102, f"{name}: {sgpr} SGPRs"
    # the default (scalar-cache) route keeps the asm body's footprint: well under the limit
    # (template arguments: K, W, VARIANT = 1, FUSED, PERSIST = 0)
    smem = [v for n, _, _, v in step if re.search(r"ELi1ELb[01]ELb0EEEvNS_10StepParamsE$", n)]
    assert len(smem) >= 14 and max(smem) <= 48, smem


def test_interaction_body_is_the_twelve_instruction_sequence(isa):
    """One (source, receiver) interaction = 2 x v_sub, v_fma, v_fmac, s_setprio, v_rsq, s_setprio, 3 x v_mul, 2 x v_fmac
    in their short encodings (the 8-byte VOP3 forms measured 11.6 % slower); with two receivers per lane the two
    interactions of a source share one priority window; and NO packed f32 instruction in the step kernels' loops
    (v_pk_add / v_pk_fma around the transcendental cost 4.5 %: DESIGN.md section 3)."""
    head = ["v_sub_f32", "v_sub_f32", "v_fma_f32", "v_fmac_f32"]
    tail = ["v_mul_f32", "v_mul_f32", "v_mul_f32", "v_fmac_f32", "v_fmac_f32"]
    single = head + ["s_setprio", "v_rsq_f32", "s_setprio"] + tail                       # K = 1
    paired = head + head + ["s_setprio", "v_rsq_f32", "v_rsq_f32", "s_setprio"] + tail + tail   # K = 2: both receivers of a lane
    for kernel, want, least in (("step_kernelILi1ELi16ELi1E", single, 16), ("step_kernelILi2ELi16ELi1E", paired, 16)):
        body = next(b for n, b in functions(isa).items() if kernel in n)
        ops = [ins.split()[0] for ins in body]
        hits = sum(1 for i in range(len(ops) - len(want) + 1) if ops[i:i + len(want)] == want)
        assert hits >= least, (kernel, hits)      # two unrolled 8-source groups at least
    assert not any(op.endswith("_e64") for op in ops if op.startswith(("v_mul_f32", "v_fmac_f32", "v_rsq_f32", "v_sub_f32")))
    # between one v_rsq_f32 and the next (one interaction's tail, the next one's head) nothing packed may appear
    rsq = [i for i, op in enumerate(ops) if op == "v_rsq_f32"]
    for a, b in zip(rsq, rsq[1:]):
        if b - a <= 14:
            assert not any(op.startswith("v_pk_") for op in ops[a:b]), ops[a:b]


def test_the_gravitational_constant_is_written_down_once(isa):
    """NB_G lives in include/nbody.h and nowhere else: the kernels get it as a launch argument from the host, like the
    reference's specialisation constant (reference src/lib/sim_gpu.c:54-72, src/shader/particle_cs.glsl:26).  No source
    under nbody_amd/csrc may spell the value, and the device code of the two kernels that multiply by G must not hold
    it as an instruction literal either."""
    header = open(os.path.join(ROOT, "include", "nbody.h")).read()
    value = float(re.search(r"^#define\s+NB_G\s+([0-9.eE+-]+)f\s*$", header, re.M).group(1))
    import struct
    literal = "0x%08x" % struct.unpack("<I", struct.pack("<f", value))[0]          # 0x41200000 for 10.0f
    spelled = re.compile(r"(?<![\w.])%s(?:\.0*)?f?(?![\w.])" % re.escape(("%g" % value)))
    csrc = os.path.join(ROOT, "nbody_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if not name.endswith((".hip", ".h", ".c")) or name in ("galaxy.c", "bench_main.c"):
            continue    # galaxy.c / bench_main.c: host code with unrelated tens (radii, warm-up steps); they use NB_G by name
        text = re.sub(r"//[^\n]*|/\*.*?\*/", "", open(os.path.join(csrc, name)).read(), flags=re.S)
        for m in spelled.finditer(text):
            line = text[text.rfind("\n", 0, m.start()) + 1:text.find("\n", m.end())]
            assert not re.search(r"f\b", m.group(0)) and "mul" not in line, f"{name}: `{line.strip()}` spells NB_G's value"
    for src in ("kernels.hip", "pipeline.hip"):
        assert "10.0f" not in open(os.path.join(csrc, src)).read(), src
    assert "NB_G" in open(os.path.join(csrc, "pipeline.hip")).read()
    fns = functions(isa)
    users = {n: b for n, b in fns.items() if "make_gm_kernel" in n or "split_sources_kernel" in n}
    assert len(users) == 2, sorted(fns)
    for name, body in users.items():
        assert any(ins.startswith("v_mul_f32") for ins in body), name
        assert not any(literal in ins.lower() or re.search(r"\b10\.0\b", ins) for ins in body), (name, literal)


def test_fused_finish_tail_uses_agent_scope_accesses_and_leaves_the_other_kernels_alone(isa):
    """The fused-finish instantiations (step_kernel<K, W, SMEM, true>) hand the parts over with agent-scope accesses: one
    8-byte `sc1` store per receiver, a wait for the workgroup's own stores before the ticket (`global_atomic_add` with
    return), sixteen 8-byte `sc1` loads issued back to back in the last arriver's tail -- and no fence (`buffer_wbl2` /
    `buffer_inv`, the L2 write-back that made round 1's first version 5-13x slower).  The unfused instantiations contain
    none of it: their code is what it was."""
    fn = functions(isa)
    # template arguments <K, W, VARIANT, FUSED, PERSIST>: ...Lb<FUSED>ELb<PERSIST>EEEv (the persistent experiment kernels,
    # PERSIST = 1, exist in TUNING=1 builds only; this is the library that ships)
    fused = {n: b for n, b in fn.items() if "step_kernel" in n and re.search(r"Lb1ELb0EEEv", n)}
    plain = {n: b for n, b in fn.items() if "step_kernel" in n and re.search(r"Lb0ELb0EEEv", n)}
    assert len(fused) == 6 and len(plain) >= 16
    assert not [n for n in fn if "step_kernel" in n and re.search(r"ELb1EEEv", n)], "persistent kernels in the shipped build"
    for name, body in fused.items():
        text = "\n".join(body)
        assert len(re.findall(r"global_store_dwordx2 .* sc1", text)) >= 1, name
        assert len(re.findall(r"global_load_dwordx2 .* sc1", text)) >= 16, name
        atom = [i for i, x in enumerate(body) if x.startswith("global_atomic_add")]
        assert len(atom) == 1 and "sc0" in body[atom[0]], name                        # the ticket, with its return value
        # order on the way to the ticket: the part store, a wait for it, the workgroup barrier, then the atomic
        store = max(i for i, x in enumerate(body[:atom[0]]) if re.match(r"global_store_dwordx2 .* sc1", x))
        wait = next(i for i in range(store + 1, atom[0]) if body[i].startswith("s_waitcnt vmcnt(0)"))
        assert any(x.startswith("s_barrier") for x in body[wait + 1:atom[0]]), name
        assert "buffer_wbl2" not in text and "buffer_inv" not in text, name
    for name, body in plain.items():
        text = "\n".join(body)
        assert "global_atomic" not in text and not re.search(r"global_(load|store)\S* .* sc1", text), name

"""CPU-side checks of the compiled gfx950 code (no GPU: hipcc cross-compiles).

1. Hand-scheduled loads. depthwise.hip (se_fc8_kernel, the squeeze-excitation FC launch) and headfuse.hip issue `global_load_*` from
   `asm volatile` statements and wait for them with hand-written `s_waitcnt vmcnt(N)` statements. LLVM believes an asm statement's output
   register is defined when the statement ends; the data arrives later. Nothing in the language stops the compiler from scheduling a copy,
   a spill or a use of such a register between the load and the wait that covers it -- so this test reads the assembly hipcc produces and
   checks, per kernel and in text order (conservative across branches): between an asm-issued VGPR load and the first hand-written wait that
   retires it (loads return in order: `vmcnt(N)` retires all but the N youngest outstanding vector-memory operations) NO instruction reads or
   writes its destination registers, and no compiler-issued vector-memory operation sits between them (it would shift the hand-made count).
2. No scratch (register spills) in the kernels whose schedules depend on counted waits or that are meant to run at a fixed occupancy.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "demonet_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


_ASM_CACHE = {}
# the flags demonet_amd/build.py compiles each file with (the hazard scan must see the code that ships)
_BUILD_EXTRA = {"pwdirect.hip": ("-mllvm", "-amdgpu-mfma-vgpr-form=1"), "pointwise.hip": ("-mllvm", "-amdgpu-mfma-vgpr-form=1"),
                "tail.hip": ("-mllvm", "-amdgpu-mfma-vgpr-form=1"), "depthwise.hip": ("-mllvm", "-amdgpu-mfma-vgpr-form=1"), "postprocess.hip": ("-ffp-contract=off",)}


def _asm_cmd(src, out, extra=()):
    return [HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fno-gpu-rdc", "--cuda-device-only", "-S", *extra, os.path.join(CSRC, src), "-o", out]


def _asm(src, tmp_path, extra=()):
    key = (src, tuple(extra))
    if key not in _ASM_CACHE:
        out = os.path.join(str(tmp_path), src.replace(".hip", ".s"))
        subprocess.check_call(_asm_cmd(src, out, extra))
        _ASM_CACHE[key] = open(out).read()
    return _ASM_CACHE[key]


def _kernels(text):
    """name -> (body lines, metadata dict) for every kernel of a device .s file"""
    res = {}
    # (up to the function-end label: a kernel with an early exit has an s_endpgm in the middle)
    for m in re.finditer(r"^(_Z\w+):\s*;\s*@\1\n(.*?)\n\.Lfunc_end\d+:", text, re.S | re.M):
        res[m.group(1)] = m.group(2).split("\n")
    meta = {}
    for m in re.finditer(r"\.amdhsa_kernel (\w+)\n(.*?)\.end_amdhsa_kernel", text, re.S):
        meta[m.group(1)] = m.group(2)
    return res, meta


def _regs(tok):
    """VGPR numbers named by one operand token: v12, v[12:15]"""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def _operands(line):
    line = line.split(";")[0].strip()
    if not line or line.endswith(":") or line.startswith("."):
        return None, []
    parts = line.split(None, 1)
    ops = [t.strip() for t in re.split(r",\s*(?![^\[]*\])", parts[1])] if len(parts) > 1 else []
    return parts[0], ops


VMEM = ("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic", "scratch_", "flat_")


def _lint_asm_loads(lines, name):
    """text-order scan; returns the number of asm-issued loads checked"""
    in_asm = False
    pending = []            # outstanding asm-issued VGPR loads, oldest first: [line number, dest registers, crossed a label]
    checked = 0
    for ln, raw in enumerate(lines):
        if "#ASMSTART" in raw:
            in_asm = True
            continue
        if "#ASMEND" in raw:
            in_asm = False
            continue
        op, ops = _operands(raw)
        if op is None:
            if raw.strip().endswith(":"):
                for q in pending:       # (labels: the scan follows the TEXT order -- a load stays 'in flight' across branch targets until a hand-written wait retires it)
                    q[2] = True
            continue
        touched = set()
        for t in ops:
            touched |= _regs(t)
        if op in ("v_mad_u64_u32", "v_mad_i64_i32") and len(ops) == 5:
            # hipcc uses the 64-bit multiply-add for 32-bit index arithmetic (only the low result half is used): the HIGH register of its addend
            # pair is then an undefined input and may be any register -- including one with a load in flight, whose value cannot reach the low
            # half. Seen in se_fc8_kernel<15,4> (v[70:71] with v71 = the bias load's destination). Not a use of the loaded value.
            hi = _regs(ops[4])
            if len(hi) == 2:
                touched -= {max(hi)} - (_regs(ops[0]) | _regs(ops[2]) | _regs(ops[3]))
        if in_asm and op.startswith(("global_load_dword", "buffer_load_dword")) and "lds" not in raw:
            pending.append([ln, _regs(ops[0]), False])
            checked += 1
            continue
        if in_asm and op.startswith(("global_store_dword", "buffer_store_dword")):
            # an asm-issued store counts in vmcnt like a load (stem_*_kernel: "the tile's own stores stay in flight"); it has no destination,
            # but a store of more than 8 bytes reads its data registers a cycle after it issues: hipcc keeps writers of those registers two
            # wait states away from the stores it knows -- for an asm statement the statement itself must (s_nop 1 behind the store)
            pending.append([ln, set(), False])
            if op.endswith(("x3", "x4")):
                data = _regs(ops[0])
                after, j = [], ln + 1
                while len(after) < 2 and j < len(lines):
                    o2, p2 = _operands(lines[j])
                    if o2 is not None:
                        after.append((o2, p2))
                    j += 1
                states = 0
                for o2, p2 in after:
                    if o2 == "s_nop":
                        states += int(p2[0]) + 1
                        continue
                    if states >= 2:
                        break
                    written = _regs(p2[0]) if p2 and not o2.startswith(("s_", "buffer_store", "global_store", "ds_write")) else set()
                    assert not (written & data), f"{name}: `{lines[j - 1].strip()}` writes the data registers of the asm-issued `{raw.strip()}` (line {ln}) inside its read window"
                    states += 1
            continue
        if in_asm and op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", raw)
            if m:
                keep = int(m.group(1))
                pending = pending[len(pending) - keep:] if keep < len(pending) else pending
            continue
        if pending:
            hot = set().union(*(q[1] for q in pending))
            assert not (touched & hot), f"{name}: `{raw.strip()}` (line {ln}) touches v{sorted(touched & hot)} while its asm-issued load is in flight"
            # a compiler-issued vector-memory operation inside a straight-line run of asm loads and their wait shifts the hand-made count
            if not in_asm and op.startswith(VMEM) and not all(q[2] for q in pending):
                raise AssertionError(f"{name}: compiler-issued `{op}` (line {ln}) between asm-issued loads and their hand-counted wait")
    left = [q for q in pending if not q[2] and q[1]]          # (stores need no wait: the end of the kernel retires them)
    assert not left, f"{name}: {len(left)} asm-issued loads never waited for"
    return checked


def test_hand_scheduled_requests_and_stores_of_the_stem_kernels(tmp_path):
    """stem_split_kernel / stem_mfma64p_kernel (depthwise.hip): the taps of the next tile(s) are requested by asm statements ahead of the matrix
    work, the tile's stores are asm statements too (a compiler-issued store between requests and wait would shift the count), and the wait is
    `vmcnt(stores issued behind the requests)`. In the compiled code: nothing touches a requested register before the wait that retires it,
    every request is waited for, no writer of a 16-byte store's data registers inside the store's read window, no scratch."""
    text = _asm("depthwise.hip", tmp_path, _BUILD_EXTRA["depthwise.hip"])
    kernels, meta = _kernels(text)
    stems = {k: v for k, v in kernels.items() if "stem_split_kernel" in k or "stem_mfma64p_kernel" in k}
    assert len(stems) >= 4, [k for k in kernels if "stem" in k]          # three split configurations + the fp32 pipeline
    for name, lines in stems.items():
        n = _lint_asm_loads(lines, name)
        assert n >= 20, (name, n)                  # 5 requests per tile x 4 or 8 tiles (14 x 8 for the fp32 kernel)
        assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", meta[name]), f"{name}: scratch in a kernel with hand-counted waits"


def test_hand_scheduled_loads_of_se_fc8_are_not_touched_before_their_wait(tmp_path):
    text = _asm("depthwise.hip", tmp_path, _BUILD_EXTRA["depthwise.hip"])
    kernels, meta = _kernels(text)
    se = {k: v for k, v in kernels.items() if "se_fc8_kernel" in k}
    assert len(se) >= 2, list(kernels)[:5]
    for name, lines in se.items():
        n = _lint_asm_loads(lines, name)
        assert n >= 20, (name, n)                  # partial rows + two FC weight batches + biases
        assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", meta[name]), f"{name}: scratch in a kernel with hand-counted waits"
    # the depthwise launches that carry the SE tail (POOL = 2) run their small-tail loads the same way
    tails = {k: v for k, v in kernels.items() if "dw_kernel" in k}
    for name, lines in tails.items():
        _lint_asm_loads(lines, name)


def test_fused_head_kernel_has_no_scratch_and_its_dma_is_counted(tmp_path):
    """head_fused_kernel: every spill reload is a vector-memory operation that waits IN ORDER behind the A-fragment requests and shifts the
    counted wait that publishes the LDS-DMA -- the kernel must compile without scratch at its 256-register budget, and every chunk iteration
    must carry the counted wait in front of its barrier."""
    text = _asm("headfuse.hip", tmp_path)
    kernels, meta = _kernels(text)
    hf = {k: v for k, v in kernels.items() if "head_fused_kernel" in k}
    assert len(hf) == 10                # TCW = 1 .. 5, each with and without the softmax / decode epilogue
    for name, lines in hf.items():
        assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", meta[name]), f"{name}: scratch"
        body = "\n".join(lines)
        assert "global_load_lds_dwordx4" in body
        tcw = int(re.search(r"head_fused_kernelILi(\d)E", name).group(1))
        assert re.search(rf"s_waitcnt vmcnt\({tcw}\)", body), f"{name}: the counted wait vmcnt({tcw}) is missing"
        assert "scratch_" not in body


def test_persistent_expdw_has_no_scratch(tmp_path):
    text = _asm("expdw.hip", tmp_path)
    kernels, meta = _kernels(text)
    one = [k for k in kernels if "expdw_one_kernel" in k]
    assert one
    for name in one:
        assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", meta[name]), f"{name}: scratch"


def test_run_staged_conv_dma_is_hidden_from_the_compiler_and_m0_is_only_touched_in_asm(tmp_path):
    """conv_halo_kernel issues its LDS-DMA (`buffer_load_dwordx4 ... lds`) from inline asm and writes M0 (the LDS address of the piece) there
    without saving it -- M0 is a reserved register that cannot be declared clobbered. That is only sound while hipcc itself never uses M0 in
    these kernels: every mention of m0 must sit inside an asm block, and no instruction that reads M0 implicitly may appear outside one. The
    stage barriers must carry the counted waits (a run piece left in flight is retired by the next stage's wait), the loop must not contain a
    compiler-issued vector-memory operation (it would shift the counts), and the kernels must not spill."""
    text = _asm("convbig.hip", tmp_path)
    kernels, meta = _kernels(text)
    halo = {k: v for k, v in kernels.items() if "conv_halo_kernel" in k}
    assert len(halo) >= 6, list(kernels)
    implicit_m0 = ("s_movrel", "v_movrel", "ds_gws", "s_sendmsg", "v_interp", "ds_add_gs", "ds_sub_gs", "s_ttrace")
    for name, lines in halo.items():
        assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", meta[name]), f"{name}: scratch"
        in_asm, dma, waits, barriers_after_first_dma = False, 0, [], 0
        last_dma_line = max(i for i, l in enumerate(lines) if "buffer_load_dwordx4" in l and " lds" in l)
        for ln, raw in enumerate(lines):
            if "#ASMSTART" in raw:
                in_asm = True
                continue
            if "#ASMEND" in raw:
                in_asm = False
                continue
            op, ops = _operands(raw)
            if op is None:
                continue
            if in_asm:
                if op.startswith("buffer_load_dwordx4"):
                    assert " lds" in raw, raw
                    dma += 1
                m = re.search(r"s_waitcnt vmcnt\((\d+)\)", raw)
                if m:
                    waits.append(int(m.group(1)))
                continue
            assert "m0" not in raw.split(";")[0], f"{name}: compiler-generated use of m0 (line {ln}): {raw.strip()}"
            assert not op.startswith(implicit_m0), f"{name}: `{op}` reads M0 implicitly (line {ln})"
            # between the first and the last DMA piece (prologue + main loop) the only vector-memory operations are the asm-issued pieces
            if dma and ln < last_dma_line:
                assert not op.startswith(VMEM), f"{name}: compiler-issued `{op}` inside the DMA-counted region (line {ln})"
        assert dma >= 40, (name, dma)                                   # prologue + 9 unrolled stages
        assert len(waits) >= 11 and waits[-1] == 0, (name, waits)       # prologue, 9 stage waits, the drain in front of the epilogue
        assert any(w > 0 for w in waits[1:-1]), (name, waits)           # counted: stages that request run pieces leave them in flight


def test_no_instruction_reads_a_matrix_result_before_it_is_written(tmp_path):
    """gfx950 does not interlock a matrix instruction's result write against a later vector / memory instruction that reads those registers:
    software keeps passes + 4 wait states between them. hipcc does that for the instructions it generates, but not for `asm` statements --
    until round 6 ReLU was an inline-asm v_max_f32 that could land 4 wait states behind the MFMA whose accumulator it read
    (tools/mfma_hazard_scan.py). Every translation unit that uses the matrix cores is scanned as it is built."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from mfma_hazard_scan import asm_hazards
    srcs = ["pointwise.hip", "pwdirect.hip", "tail.hip", "expdw.hip", "headfuse.hip", "convbig.hip"]
    procs = []
    for src in srcs:
        key = (src, tuple(_BUILD_EXTRA.get(src, ())))
        if key in _ASM_CACHE:
            continue
        out = os.path.join(str(tmp_path), "hz_" + src.replace(".hip", ".s"))
        procs.append((key, out, subprocess.Popen(_asm_cmd(src, out, key[1]))))
    for key, out, p in procs:
        assert p.wait() == 0, key
        _ASM_CACHE[key] = open(out).read()
    for src in srcs:
        text = _ASM_CACHE[(src, tuple(_BUILD_EXTRA.get(src, ())))]
        assert "v_mfma" in text, src
        bad = asm_hazards(text)        # (consumers in `asm` statements: the ones hipcc's own hazard recognizer cannot see)
        assert not bad, f"{src}: {len(bad)} inline-asm reads of a matrix result inside its hazard window, first: {bad[0]}"


def test_every_asm_issued_wide_store_carries_its_wait_state():
    """A store of more than 8 bytes reads its data registers a cycle after it issues; hipcc keeps writers of those registers away from the stores IT
    emits, an `asm` statement is opaque to it (round 6: 0.2 % of stem_split_kernel's first outputs were non-deterministic). Source-level rule for
    every kernel file: an asm statement that issues a 12- or 16-byte store ends with `s_nop 1` (or more) inside the same statement."""
    pat = re.compile(r'asm\s+volatile\s*\(\s*"((?:[^"\\]|\\.)*)"')
    seen = 0
    for fn in sorted(os.listdir(CSRC)):
        if not fn.endswith((".hip", ".h")):
            continue
        for m in pat.finditer(open(os.path.join(CSRC, fn)).read()):
            text = m.group(1)
            if re.search(r"(buffer|global|flat|scratch)_store_dwordx[34]", text):
                seen += 1
                assert re.search(r"store_dwordx[34][^\\]*\\n\\ts_nop [1-9]", text), f"{fn}: `{text}` has no wait state behind its store"
    assert seen >= 2


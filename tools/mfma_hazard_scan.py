"""dev tool + test helper: finds MFMA -> VALU / memory read-after-write hazards that the compiler could not see.

gfx950 has no hardware interlock between a matrix instruction's result write and a later NON-matrix instruction that reads (or overwrites)
those registers: software must keep NumPasses + 4 wait states between them (v_mfma_*_32x32x16: 8 passes -> 12; v_mfma_*_16x16x32: 4 -> 8;
fp32-input forms 32x32x2: 16 -> 20, 16x16x4: 8 -> 12; LLVM GCNHazardRecognizer, GFX940_XDL_N_PassWriteVgprVALUMemExpReadWaitStates).
hipcc inserts the `s_nop`s itself for every instruction it knows -- but an `asm volatile("v_max_f32 ...")` statement is opaque to its
hazard recognizer, so an inline-asm VALU instruction that consumes an accumulator can be scheduled right behind the MFMA and read stale
registers (round 6: found in the lab copy of the 1x1 kernel with a ReLU epilogue).

    python tools/mfma_hazard_scan.py file.s [...]         (device assembly from hipcc -S --cuda-device-only)

Two classes of findings. (1) The consumer sits in an `asm` statement: the compiler did not -- could not -- protect it; the lint test fails on these.
(2) The consumer is compiler-generated: hipcc's own recognizer counted enough wait states on the path IT walked; this scanner walks every forward
path and finds shorter ones across multi-predecessor blocks (the activation switch behind an MFMA: 5 - 7 wait states through two or three scalar
branches where the straight path gets its full `s_nop 11`). Those are reported (`--all`), not failed: every taken branch refills the instruction
buffer, which is far more than the missing wait states, and the same pattern has been in every kernel of this library since round 2.

Text order with forward control flow followed: a forward branch carries the open windows (with the wait states elapsed so far) to its
target label, where they are merged with the fall-through state by the smaller elapsed count. Backward branches (loops) are not followed:
a window never stays open across a loop's back edge in these kernels (every loop body ends in stores or a barrier far behind its MFMAs).
"""
import re
import sys

PASSES = [
    (re.compile(r"v_mfma_\w+_32x32x2_?f32|v_mfma_f32_32x32x2f32"), 16),
    (re.compile(r"v_mfma_\w+_16x16x4_?f32|v_mfma_f32_16x16x4f32"), 8),
    (re.compile(r"v_mfma_\w+_32x32x(16|8)_"), 8),          # gfx950 double-K forms and the legacy forms priced alike (conservative: 8 passes -> 12)
    (re.compile(r"v_mfma_\w+_16x16x(32|16)_"), 4),
    (re.compile(r"v_mfma_"), 16),                           # anything else: assume the longest
]


def _regs(tok):
    tok = tok.strip()
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def _split(line):
    line = line.split(";")[0].strip()
    if not line or line.endswith(":") or line.startswith(".") or line.startswith("//"):
        return None, []
    parts = line.split(None, 1)
    ops = [t.strip() for t in re.split(r",\s*(?![^\[]*\])", parts[1])] if len(parts) > 1 else []
    return parts[0], ops


def scan(text):
    """-> list of (kernel, line number, mfma text, consumer text, wait states seen, wait states needed)"""
    bad = []
    kernel = None
    live = []          # [regs, needed, elapsed, mfma_line, id]
    pending = {}       # label -> windows carried by forward branches
    fallthrough = True
    in_asm = False

    def merge(a, b):
        by = {}
        for e in a + b:
            if e[4] not in by or e[2] < by[e[4]][2]:
                by[e[4]] = list(e)
        return list(by.values())
    for ln, raw in enumerate(text.split("\n"), 1):
        m = re.match(r"^(_Z\w+|\w+):\s*;\s*@", raw)
        if m:
            kernel, live, pending, fallthrough = m.group(1), [], {}, True
            continue
        s = raw.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        ml = re.match(r"^(\.?LBB\w+):", s)
        if ml:
            live = merge(live if fallthrough else [], pending.pop(ml.group(1), []))
            fallthrough = True
            continue
        op, ops = _split(raw)
        if op is None:
            continue
        if op == "s_endpgm" or op.startswith("s_setpc"):
            live, fallthrough = [], False
            continue
        if op.startswith("s_cbranch") or op.startswith("s_branch"):
            for e in live:
                e[2] += 1
            live = [e for e in live if e[2] < e[1]]
            if ops:
                pending[ops[0]] = merge(pending.get(ops[0], []), [list(e) for e in live])
            if op.startswith("s_branch"):
                live, fallthrough = [], False
            continue
        ws = 1
        if op == "s_nop":
            ws = int(ops[0], 0) + 1
        if op.startswith("v_mfma") or op.startswith("v_smfmac"):
            # the matrix pipe orders dependent matrix instructions itself (the compiler adds what is needed); it starts a new window
            for e in live:
                e[2] += ws
            passes = next(p for rx, p in PASSES if rx.search(op))
            live = [e for e in live if e[2] < e[1]]
            live.append([_regs(ops[0]), passes + 4, 0, raw.strip(), ln])
            continue
        touched = set()
        for t in ops:
            touched |= _regs(t)
        is_vec = op.startswith(("v_", "ds_", "global_", "buffer_", "flat_", "scratch_", "exp"))
        if is_vec and touched:
            for e in live:
                if e[2] < e[1] and (touched & e[0]):
                    bad.append((kernel, ln, e[3], raw.strip() + ("   [inline asm]" if in_asm else ""), e[2], e[1]))
        for e in live:
            e[2] += ws
        live = [e for e in live if e[2] < e[1]]
    return bad


def asm_hazards(text):
    """the findings whose consumer is an inline-asm statement (what the lint test asserts on)"""
    return [b for b in scan(text) if b[3].endswith("[inline asm]")]


if __name__ == "__main__":
    show_all = "--all" in sys.argv
    total = other = 0
    for path in [a for a in sys.argv[1:] if not a.startswith("--")]:
        for k, ln, mf, use, seen, need in scan(open(path).read()):
            if use.endswith("[inline asm]"):
                total += 1
            else:
                other += 1
                if not show_all:
                    continue
            print(f"{path}:{ln}: {k}\n    {mf}\n    -> {use}\n    {seen} wait states between them, {need} needed")
    print(f"{total} inline-asm hazards; {other} compiler-generated consumers on a forward path shorter than the table's wait states (--all lists them)")
    sys.exit(1 if total else 0)

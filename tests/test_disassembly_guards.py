"""Build-time guards on the kernels whose LDS-DMA / asm-issued loads are released by HAND-COUNTED ``s_waitcnt vmcnt(N)``:
the counts assume that every vector-memory instruction inside the counted regions is one the source issues.  A register spill the
compiler adds there (``scratch_load`` / ``scratch_store`` are vector-memory operations too) would not make a wait too short -- the
operations waited for are always OLDER than the spill traffic -- but it would make the compiler put its own ``vmcnt(0)`` in front of the
reload's first use, i.e. drain the DMA ring, and it is the first symptom of a changed register allocation.  So: disassemble the code
object inside libdkd.so and require the loops of those kernels to be free of scratch traffic.  CPU-only (llvm-objdump of the built library).
"""
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(ROOT, "deltakd_amd", "lib", "libdkd.so")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


@pytest.fixture(scope="module")
def kernels():
    """{mangled kernel name: [instruction lines]} for every kernel of libdkd.so's gfx950 code objects."""
    if not (os.path.exists(LIB) and os.path.exists(os.path.join(LLVM, "llvm-objdump"))):
        pytest.skip("libdkd.so or llvm-objdump not available")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", LIB], check=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        assert starts, "no offload bundle in libdkd.so"
        for i, st in enumerate(starts):                     # one bundle per translation unit
            part = os.path.join(tmp, f"b{i}.bin")
            open(part, "wb").write(blob[st:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = os.path.join(tmp, f"b{i}.co")
            r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={part}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout
            cur = None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
                if m:
                    cur = m.group(1)
                    out[cur] = []
                elif cur is not None and line.strip():
                    out[cur].append(line.strip())
    return out


def _body(lines):
    return [ln.split("//")[0].strip() for ln in lines]


def _find(kernels, *needles):
    hits = [k for k in kernels if all(n in k for n in needles)]
    assert hits, (needles, sorted(kernels)[:5])
    return hits


@pytest.mark.parametrize("needles,what", [
    (("attn192_bwd_kernel", "ILb0E"), "fused proj dgrad + attention backward (default instantiation)"),
    (("attn192_fwd_kernel", "ILi12E"), "fused qkv + attention forward (the 197 / 198-token instantiation; the generic-N one may spill)"),
    (("mlp192_kernel",), "fused MLP forward / backward"),
])
def test_counted_vmcnt_kernels_have_no_spill_traffic_in_their_loops(kernels, needles, what):
    """Between the kernel's first workgroup barrier and its end -- the persistent loop with the LDS-DMA rings -- there must be no scratch
    (spill) instruction and no flat_ load (flat operations complete out of order: counted vmcnt waits do not cover them)."""
    for name in _find(kernels, *needles):
        body = _body(kernels[name])
        first = next((i for i, ln in enumerate(body) if ln.startswith("s_barrier")), None)
        assert first is not None, (name, "no barrier?")
        loop = body[first:]
        bad = [ln for ln in loop if ln.startswith(("scratch_", "flat_load", "flat_store"))]
        assert not bad, f"{what} ({name}): {len(bad)} spill / flat instructions inside the counted-wait region, e.g. {bad[:3]}"
        assert any(ln.startswith("global_load_lds_dwordx4") for ln in loop), (name, "the LDS-DMA pieces should be in this region")


def test_attn192_bwd_counted_waits_match_the_issue_pattern(kernels):
    """dkd_attn192_bwd (default instantiation): the counted wait on a head's q / k / v / O pieces is there with the counts the piece
    distribution gives for 197 / 198 tokens (vmcnt(12) and vmcnt(13): 100 pieces over 8 waves), nothing waits with a LARGER count (that would
    be a wait that assumes more traffic in flight than the source issues), and the only asm-issued vector-memory instructions are
    LDS-DMA pieces: every global load with a register result is the compiler's own (it is followed by the compiler's own wait)."""
    for name in _find(kernels, "attn192_bwd_kernel", "ILb0E"):
        body = _body(kernels[name])
        waits = {int(m.group(1)) for ln in body for m in [re.match(r"s_waitcnt vmcnt\((\d+)\)", ln)] if m}
        waits.discard(63)                                   # (how the disassembler prints "no wait on this counter")
        assert {12, 13} <= waits and max(waits) == 13, (name, sorted(waits))
        assert sum(ln.startswith("global_load_lds_dwordx4") for ln in body) >= 4, name


def _regs_of(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def _reads_of_in_flight_registers(lines):
    """Linear scan of a kernel's instructions: the destination of a global / scratch load is "in flight" until an ``s_waitcnt vmcnt(N)``
    retires it (all but the N youngest vector-memory operations) or the register is written again; any instruction that READS such a
    register in between is returned.  For compiler-issued loads this can never happen (the compiler waits before the first use); it
    happens when an asm-issued load's destination is copied / spilled by the compiler before the asm statement that waits for it."""
    inflight, bad = [], []
    for ln in lines:
        parts = ln.replace(",", " ").split()
        if not parts:
            continue
        op, args = parts[0], parts[1:]
        if op == "s_waitcnt":
            m = next((re.fullmatch(r"vmcnt\((\d+)\)", a) for a in args if a.startswith("vmcnt")), None)
            if m:
                n = int(m.group(1))
                inflight = inflight[len(inflight) - n:] if n else []
            continue
        is_store = op.startswith(("global_store", "scratch_store", "ds_write", "ds_add", "global_atomic", "buffer_store"))
        pending = set().union(*[d for d in inflight if d]) if inflight else set()
        for a in (args if is_store else args[1:]):
            hit = _regs_of(a) & pending
            if hit:
                bad.append((ln, sorted(hit)))
        if not is_store and args:
            w = _regs_of(args[0])
            if w:
                inflight = [None if d is None else (d - w) for d in inflight]
        if op.startswith(("global_load", "global_store", "global_atomic", "buffer_", "scratch_", "flat_")):
            if op.startswith(("global_load_lds", "buffer_load_lds")) or is_store or not op.startswith(("global_load", "scratch_load", "buffer_load", "flat_load")):
                inflight.append(None)
            else:
                inflight.append(_regs_of(args[0]))
            inflight = inflight[-64:]
    return bad


@pytest.mark.parametrize("needles", [("attn192_bwd_kernel", "ILb0E"), ("attn192_bwd_kernel", "ILb1E"), ("attn192_bwd_phase_c",),
                                     ("attn192_fwd_kernel", "ILi12E"), ("mlp192_kernel",)])
def test_no_instruction_reads_a_register_whose_load_is_still_in_flight(kernels, needles):
    """Round 4 bug (found by poisoning LDS / VGPRs with NaNs, tests/test_attn192_gpu.py::test_attn192_bwd_does_not_read_lds_it_never_wrote):
    dkd_attn192_bwd issues some of its global loads from asm and releases them with a later, counted ``s_waitcnt`` statement tied to the
    destination registers.  When compiler-visible control flow (``if (nq == 10) wait<10>(..) else if ..``) sat between issue and wait,
    hipcc satisfied each branch's ties with ``v_mov`` copies of the destinations IN FRONT of the wait -- copies of registers whose data had
    not arrived.  The waits now branch inside one asm statement; this test keeps it that way for every kernel that uses the idiom."""
    hits = [k for k in kernels if all(n in k for n in needles)]
    if not hits and needles == ("attn192_bwd_phase_c",):
        pytest.skip("phase C was inlined")
    assert hits, needles
    for name in hits:
        bad = _reads_of_in_flight_registers(_body(kernels[name]))
        assert not bad, f"{name}: {len(bad)} reads of in-flight load destinations, e.g. {bad[:4]}"

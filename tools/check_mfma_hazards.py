"""Static half of the kernel reproducibility check: scan the disassembly of the hand-scheduled kernels
for a VALU / VMEM / LDS instruction that READS a register an MFMA wrote fewer than the required wait states earlier.

hipcc pads these hazards for code it schedules itself but NOT inside an inline-asm statement (the round-4 attention bug:
an asm v_max3_f32 read score accumulators up to 12 states early).  Rule checked (gfx950, cdna_hip_programming.md 5.7 item 2):
after v_mfma_f32_32x32x16_bf16 / 32x32x2_f32 (8 / 16 passes) a non-MFMA reader of its destination needs >= 12 (bf16 form)
or >= 18 (f32 32x32x2: 16-pass) wait states; the next MFMA taking it whole as C needs none; an overlapping-but-different
C or an A / B read needs the full distance.  Every instruction issues >= 1 state, `s_nop N` N + 1; we count conservatively
(1 per instruction, MFMAs 1: the real issue cost is higher), so a report here means "look", not "broken".

Second rule (round 5, the dK/dV kernel's hand-issued mask loads): a scalar load writes its destination whenever it returns,
so between an `s_load_*` and the next `s_waitcnt lgkmcnt(0)` NOTHING may read or write its destination SGPRs - a destination
the compiler believes dead (an unused asm output) gets handed to something else and is then overwritten behind its back; one
it copies or spills early holds stale data.  hipcc keeps this for its own loads; the asm-issued ones are checked here.

usage: check_mfma_hazards.py [libtethys_mi.so]   (exit code 1 if a candidate is found)"""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tethys-speech_amd", "libtethys_mi.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
NEED = {"v_mfma_f32_32x32x16_bf16": 12, "v_mfma_f32_32x32x2_f32": 18, "v_mfma_f32_16x16x32_bf16": 8, "v_mfma_f32_16x16x4_f32": 10}
KERNELS = re.compile(r"attn_(fwd|bwd_dq|bwd_dkv|bwd_small)_kernel|gemm_p8_kernel|gemm_fast_kernel|gemm_f32_kernel")


def regs(tok):
    """register operand text -> set of (file, index)"""
    out = set()
    for m in re.finditer(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b", tok):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def extract_bundles(path):
    """device code objects embedded in the host library (llvm-objdump --offloading writes them next to its input: a copy)"""
    import shutil
    tmp = "/tmp/_tmi_hazard"
    shutil.rmtree(tmp, ignore_errors=True)
    os.makedirs(tmp)
    cp = os.path.join(tmp, "lib.so")
    shutil.copy(path, cp)
    subprocess.run([OBJDUMP, "--offloading", cp], capture_output=True, text=True, cwd=tmp)
    return sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if f.endswith("gfx950"))


def scan(asm_lines):
    bad = []
    kernel, track = None, []   # track: list of [regs, need, age, mfma text]
    for line in asm_lines:
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            kernel = m.group(1) if KERNELS.search(m.group(1)) else None
            track = []
            continue
        if kernel is None:
            continue
        ins = line.split("//")[0].strip()
        if not ins or ins.startswith("s_code_end"):
            continue
        parts = ins.split(None, 1)
        op, args = parts[0], (parts[1] if len(parts) > 1 else "")
        ops_ = [a.strip() for a in args.split(",")]
        states = 1
        if op == "s_nop":
            states = int(ops_[0], 0) + 1 if ops_ and ops_[0] else 1
        if op.startswith("v_mfma"):
            dst, srcs = regs(ops_[0]), [regs(o) for o in ops_[1:4]]
            for t in track:
                if t[2] < t[1]:
                    # whole-C accumulate chaining is free; any other overlap inside the window is a candidate
                    if srcs and len(srcs) == 3 and srcs[2] == t[0]:
                        continue
                    if any(s & t[0] for s in srcs) or (dst & t[0] and dst != t[0]):
                        bad.append((kernel, t[3], ins, t[2], t[1]))
            for t in track:
                t[2] += states
            base = op.split("_e64")[0]
            track = [t for t in track if t[2] < t[1]]
            track.append([dst, NEED.get(base, 18), 0, ins])
            continue
        if op.startswith(("v_", "ds_", "global_", "buffer_", "flat_", "scratch_")):
            read = set()
            for o in (ops_[1:] if op.startswith("v_") and not op.startswith("v_cmp") else ops_):
                read |= regs(o)
            if op.startswith(("ds_write", "global_store", "buffer_store", "scratch_store", "global_atomic", "ds_bpermute", "ds_swizzle")) or op.startswith("v_cmp"):
                read |= regs(args)
            for t in track:
                if t[2] < t[1] and read & t[0]:
                    bad.append((kernel, t[3], ins, t[2], t[1]))
        for t in track:
            t[2] += states
        track = [t for t in track if t[2] < t[1]]
        if op in ("s_branch", "s_endpgm", "s_setpc_b64", "s_swappc_b64"):
            track = []  # the text that follows is not reached by falling through: another path's code
        # (a conditional branch keeps the window open on its fall-through side; its taken side only adds states)
    return bad


def sregs(tok):
    out = set()
    for m in re.finditer(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b", tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def scan_scalar_loads(asm_lines):
    """[(kernel, load, offending instruction)]: an SGPR touched while a scalar load into it may still be in flight."""
    bad = []
    kernel, flight = None, []   # flight: [(dest regs, load text)]
    for line in asm_lines:
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            kernel = m.group(1) if KERNELS.search(m.group(1)) else None
            flight = []
            continue
        if kernel is None:
            continue
        ins = line.split("//")[0].strip()
        if not ins:
            continue
        parts = ins.split(None, 1)
        op, args = parts[0], (parts[1] if len(parts) > 1 else "")
        if op == "s_waitcnt":
            if "lgkmcnt(0)" in args:
                flight = []
            continue
        touched = sregs(args)
        for dst, text in flight:
            if touched & dst:
                bad.append((kernel, text, ins))
        if op.startswith("s_load_") or op.startswith("s_buffer_load_"):
            flight.append((sregs(args.split(",")[0]), ins))
        if op in ("s_branch", "s_endpgm", "s_setpc_b64", "s_swappc_b64"):
            flight = []
    return bad


def main():
    objs = extract_bundles(lib)
    if not objs:
        print("no gfx950 code object found in", lib)
        return 2
    total, sl = [], []
    for o in objs:
        asm = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", o], capture_output=True, text=True).stdout.split("\n")
        total += scan(asm)
        sl += scan_scalar_loads(asm)
    seen_sl = set()
    for k, ld, ins in sl:
        key = (k[:60], ld, ins)
        if key not in seen_sl:
            seen_sl.add(key)
            print(f"{k[:70]}\n    {ld}\n    destination touched before lgkmcnt(0): {ins}")
    print(f"{len(seen_sl)} scalar-load destination(s) touched in flight")
    seen = set()
    for k, mf, ins, age, need in total:
        key = (k[:60], mf, ins)
        if key in seen:
            continue
        seen.add(key)
        print(f"{k[:70]}\n    {mf}\n    read {age} states later (needs {need}): {ins}")
    print(f"{len(seen)} candidate hazard(s) in {len(objs)} code object(s)")
    return 1 if (seen or seen_sl) else 0


if __name__ == "__main__":
    sys.exit(main())

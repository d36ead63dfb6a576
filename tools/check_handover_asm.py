#!/usr/bin/env python3
"""Hold the COMPILED step train to its hand-over protocol (csrc/rem2d_vel4.h, rem2d_step_train_kernel*).

The step train hands a block's state from the workgroup of step s to the workgroup of step s + 1 through the XCD's L2 and one
flag word.  Its correctness rests on two instruction sequences that no fence of the language spells out on gfx950 (a
workgroup-scope release compiles to nothing, round 5's finding), so they are inline assembly -- and this script reads them
back from the code object of a built library:

  publish   walking BACKWARDS from the flag store (`global_store_dword ... sc1`) through the kernel's control-flow graph, every
            path meets an `s_waitcnt vmcnt(0)` before it meets a store / atomic to memory: every store of the item has been
            acknowledged by the L2 before the flag can be seen.  Nothing is stored to memory after the flag.
  acquire   whatever runs after a poll of the flag (`global_load_dword ... sc1`) meets `buffer_inv sc1`, `s_dcache_inv`,
            `s_waitcnt vmcnt(0) lgkmcnt(0)` (back to back) before any other vector load of memory, and no such load sits on
            a path from the kernel's entry to that invalidate either: the CU's L1 and the scalar cache are dropped before any
            state of the block is read.

Not judged: scalar loads (s_load_*) inside the poll section.  The ones there read the kernel-argument segment (a constant: e.g. the
failure counter's address), and everything scalar loaded AFTER the section is behind its `s_dcache_inv`.

The graph is read from the disassembly: direct branches, and LLVM's long-branch expansion (s_getpc / s_add / s_addc /
s_setpc) folded into direct ones; any other computed jump is refused.

Also reported (not judged): the kernel's registers, spills, scratch and LDS from the code object's metadata -- the figures
DESIGN.md quotes.  Runs anywhere (no GPU): `python tools/check_handover_asm.py [lib.so ...]`; imported by
__graft_entry__.build() and tests/test_abi.py.  Test infrastructure: nothing in the product path imports it.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRAIN = re.compile(r"^_Z\d+rem2d_step_train\w*_kernel")   # every instantiation of the train


class HandoverAsmError(AssertionError):
    pass


def _run(*cmd):
    return subprocess.check_output(list(cmd), stderr=subprocess.STDOUT).decode("utf-8", "replace")


def code_object(lib_path, workdir):
    """The gfx950 code object inside a HIP shared library (section .hip_fatbin -> clang-offload-bundler)."""
    fat = os.path.join(workdir, "fat.bin")
    co = os.path.join(workdir, "dev.co")
    _run(LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib_path, fat)
    _run(LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
         "--input=" + fat, "--output=" + co)
    return co


def train_symbols(co):
    out = _run(LLVM + "/llvm-readelf", "-s", "--wide", co)
    syms = []
    for line in out.splitlines():
        f = line.split()
        if len(f) >= 8 and f[3] == "FUNC" and TRAIN.match(f[7]):
            syms.append(f[7])
    return sorted(set(syms))


def metadata(co, symbol):
    """vgprs / spills / scratch / LDS of one kernel from the AMDGPU metadata note."""
    notes = _run(LLVM + "/llvm-readelf", "--notes", co)
    i = notes.find(".name:           " + symbol + "\n")
    if i < 0:
        return {}
    # a kernel's entry runs from the previous "  - .agpr_count" to the next one
    lo = notes.rfind("  - .agpr_count", 0, i)
    hi = notes.find("  - .agpr_count", i)
    blk = notes[lo:hi if hi > 0 else len(notes)]
    keys = {"vgprs": "vgpr_count", "sgprs": "sgpr_count", "vgpr_spills": "vgpr_spill_count", "sgpr_spills": "sgpr_spill_count",
            "scratch_bytes_per_lane": "private_segment_fixed_size", "lds_bytes": "group_segment_fixed_size",
            "kernarg_bytes": "kernarg_segment_size"}
    out = {}
    for k, name in keys.items():
        m = re.search(r"\.%s:\s+(\d+)" % name, blk)
        if m:
            out[k] = int(m.group(1))
    return out


_INS = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):")
_TGT = re.compile(r"<[^>+]+\+0x([0-9A-Fa-f]+)>")


def _imm(op):
    op = op.strip()
    return int(op, 16) if op.lower().startswith(("0x", "-0x")) else int(op)


def disassemble(co, symbol):
    """[(address, mnemonic, operands, branch target address or None)] of one kernel.  LLVM's long-branch expansion
    (`s_getpc_b64 s[a:b]; s_add_u32 sa, sa, lo; s_addc_u32 sb, sb, hi; s_setpc_b64 s[a:b]`) is folded into a direct branch at the
    s_setpc; any other s_setpc / s_swappc is refused (the checks below follow direct control flow only)."""
    txt = _run(LLVM + "/llvm-objdump", "-d", "--disassemble-symbols=" + symbol, co)
    base = None
    ins = []
    for line in txt.splitlines():
        m = re.match(r"^([0-9a-f]+) <%s>:" % re.escape(symbol), line)
        if m:
            base = int(m.group(1), 16)
            continue
        m = _INS.match(line)
        if not m:
            continue
        mnem, ops, addr = m.group(1), m.group(2), int(m.group(3), 16)
        tgt = None
        if mnem.startswith("s_cbranch") or mnem == "s_branch":
            t = _TGT.search(line)
            if t is None:
                raise HandoverAsmError("%s: cannot read the target of `%s %s`" % (symbol, mnem, ops))
            tgt = base + int(t.group(1), 16)
        elif mnem in ("s_setpc_b64", "s_swappc_b64", "s_call_b64"):
            ok = mnem == "s_setpc_b64" and len(ins) >= 3 and ins[-3][1] == "s_getpc_b64" and ins[-2][1] == "s_add_u32" \
                and ins[-1][1] == "s_addc_u32" and ins[-3][2].strip() == ops.strip()
            if not ok:
                raise HandoverAsmError("%s: indirect control flow (%s at %#x): the check cannot follow it" % (symbol, mnem, addr))
            lo = _imm(ins[-2][2].split(",")[-1]) & 0xffffffff
            hi = _imm(ins[-1][2].split(",")[-1]) & 0xffffffff
            off = (hi << 32) | lo
            if off >= 1 << 63:
                off -= 1 << 64
            tgt = ins[-3][0] + 4 + off   # (s_getpc_b64 returns the address of the instruction after it)
        ins.append((addr, mnem, ops, tgt))
    if base is None or not ins:
        raise HandoverAsmError("no disassembly for " + symbol)
    return ins


_MEM_STORE = re.compile(r"^(global|flat|buffer)_(store|atomic)")   # stores that leave the CU (scratch / LDS are the wavefront's own)
_VLOAD = re.compile(r"^(global|flat|buffer)_load")


def _cfg(ins):
    """successors / predecessors per instruction index (direct control flow)."""
    index = {a: k for k, (a, m, o, t) in enumerate(ins)}
    succ = [[] for _ in ins]
    for k, (a, m, o, t) in enumerate(ins):
        if m == "s_endpgm":
            continue
        if t is not None:
            if t not in index:
                raise HandoverAsmError("branch at %#x leaves the kernel (target %#x)" % (a, t))
            succ[k].append(index[t])
            if m == "s_branch" or m == "s_setpc_b64":
                continue
        if k + 1 < len(ins):
            succ[k].append(k + 1)
    pred = [[] for _ in ins]
    for k, ss in enumerate(succ):
        for j in ss:
            pred[j].append(k)
    return succ, pred


def _reach(start, edges, stop):
    """indices reachable from `start` (a list) along `edges` without expanding the nodes for which stop(k) holds (they are
    included, their neighbours are not)."""
    seen = set(start)
    todo = list(start)
    while todo:
        k = todo.pop()
        if stop(k):
            continue
        for j in edges[k]:
            if j not in seen:
                seen.add(j)
                todo.append(j)
    return seen


def check_kernel(ins, symbol):
    """Raise HandoverAsmError unless the publish and acquire sequences hold on EVERY path of the kernel's control-flow graph;
    return what was found."""
    succ, pred = _cfg(ins)
    is_flag = lambda k: ins[k][1] == "global_store_dword" and re.search(r"\bsc1\b", ins[k][2]) is not None
    is_poll = lambda k: ins[k][1] == "global_load_dword" and re.search(r"\bsc1\b", ins[k][2]) is not None
    is_wait0 = lambda k: ins[k][1] == "s_waitcnt" and "vmcnt(0)" in ins[k][2]
    is_inv = lambda k: ins[k][1] == "buffer_inv"

    # ---- acquire ----
    polls = [k for k in range(len(ins)) if is_poll(k)]
    invs = [k for k in range(len(ins)) if is_inv(k)]
    if not polls:
        raise HandoverAsmError("%s: no agent-scope poll of the flag (global_load_dword ... sc1) found" % symbol)
    if len(invs) != 1:
        raise HandoverAsmError("%s: expected exactly one buffer_inv, found %d" % (symbol, len(invs)))
    k_inv = invs[0]
    if not re.search(r"\bsc1\b", ins[k_inv][2]):
        raise HandoverAsmError("%s: buffer_inv without sc1" % symbol)
    seq = [(ins[k_inv + j][1], ins[k_inv + j][2]) for j in (1, 2)]
    if seq[0][0] != "s_dcache_inv" or seq[1][0] != "s_waitcnt" or "vmcnt(0)" not in seq[1][1] or "lgkmcnt(0)" not in seq[1][1]:
        raise HandoverAsmError("%s: buffer_inv sc1 is not followed by s_dcache_inv + s_waitcnt vmcnt(0) lgkmcnt(0): %r" % (symbol, seq))
    # (a) whatever runs after a poll meets the invalidate before it loads anything else from memory (or it ends)
    after_poll = _reach(polls, succ, is_inv)
    for k in sorted(after_poll):
        if _VLOAD.match(ins[k][1]) and not is_poll(k):
            raise HandoverAsmError("%s: `%s %s` at %#x can run after a poll of the flag and before the invalidate at %#x"
                                   % (symbol, ins[k][1], ins[k][2], ins[k][0], ins[k_inv][0]))
    # (b) nothing is loaded from memory on the way from the kernel's entry to the invalidate either (a load hoisted above the wait
    #     would hold the previous step's data); paths that never reach the invalidate (step 0 of a launch) may load what they like
    from_entry = _reach([0], succ, is_inv)
    to_inv = _reach([k_inv], pred, lambda k: False)
    for k in sorted(from_entry & to_inv):
        if _VLOAD.match(ins[k][1]) and not is_poll(k):
            raise HandoverAsmError("%s: `%s %s` at %#x loads memory on the way to the invalidate at %#x"
                                   % (symbol, ins[k][1], ins[k][2], ins[k][0], ins[k_inv][0]))

    # ---- publish ----
    # the flag of the block: the sc1 store that is NOT part of the poll section (there: flags[1] of a wait that ran into its limit)
    flag_stores = [k for k in range(len(ins)) if is_flag(k) and k not in (from_entry & to_inv)]
    if len(flag_stores) != 1:
        raise HandoverAsmError("%s: expected exactly one publishing flag store (global_store_dword ... sc1), found %d"
                               % (symbol, len(flag_stores)))
    k_flag = flag_stores[0]
    # walking BACKWARDS from the flag store, every path meets an `s_waitcnt vmcnt(0)` before it meets a store to memory
    back = _reach(pred[k_flag], pred, is_wait0)
    for k in sorted(back):
        if _MEM_STORE.match(ins[k][1]):
            raise HandoverAsmError("%s: `%s %s` at %#x can run after the last `s_waitcnt vmcnt(0)` and before the flag store at %#x -- "
                                   "the flag could reach the L2 before it" % (symbol, ins[k][1], ins[k][2], ins[k][0], ins[k_flag][0]))
    if 0 in back and not is_wait0(0):
        raise HandoverAsmError("%s: a path from the kernel's entry reaches the flag store without any `s_waitcnt vmcnt(0)`" % symbol)
    waits = sorted(k for k in back if is_wait0(k))
    # nothing is stored to memory after the flag either (the next step may already be reading)
    after_flag = _reach(succ[k_flag], succ, lambda k: False)
    for k in sorted(after_flag):
        if _MEM_STORE.match(ins[k][1]):
            raise HandoverAsmError("%s: `%s %s` at %#x can run after the flag store" % (symbol, ins[k][1], ins[k][2], ins[k][0]))
    return {"flag_store": "%#x" % ins[k_flag][0], "publish_waitcnt": ["%#x" % ins[k][0] for k in waits],
            "instructions_between": len(back) - len(waits), "polls": len(polls), "invalidate": "%#x" % ins[k_inv][0],
            "instructions": len(ins), "long_branches": sum(1 for i in ins if i[1] == "s_setpc_b64")}


def check_library(lib_path):
    """{kernel symbol: {publish / acquire findings, resources}} for every step train kernel of one library."""
    with tempfile.TemporaryDirectory() as wd:
        co = code_object(lib_path, wd)
        syms = train_symbols(co)
        if not syms:
            raise HandoverAsmError("%s: no rem2d_step_train*_kernel in the gfx950 code object" % lib_path)
        out = {}
        for sym in syms:
            r = check_kernel(disassemble(co, sym), sym)
            r["resources"] = metadata(co, sym)
            out[sym] = r
        return out


def scratch_map(lib_path, bins=40):
    """Where a train kernel's register spills sit: per 1/bins-th of its instructions (address order) the scratch stores / loads,
    SGPR-spill lane moves, LDS and global memory instructions.  The layout of the step train in address order is: `pre` (six
    lane-count instantiations), the velocity tile (one), then `post` + the TOI solve (six instantiations)."""
    with tempfile.TemporaryDirectory() as wd:
        co = code_object(lib_path, wd)
        out = {}
        for sym in train_symbols(co):
            ins = disassemble(co, sym)
            rows = [[0, 0, 0, 0, 0, 0] for _ in range(bins)]
            for k, (a, m, o, t) in enumerate(ins):
                r = rows[k * bins // len(ins)]
                r[0] += 1
                r[1] += m.startswith("scratch_store")
                r[2] += m.startswith("scratch_load")
                r[3] += m in ("v_writelane_b32", "v_readlane_b32")
                r[4] += m.startswith("ds_")
                r[5] += m.startswith("global_")
            out[sym] = rows
        return out


def default_libraries():
    pkg = os.path.join(ROOT, "gym_rem2d_amd")
    return [os.path.join(pkg, n) for n in ("librem2d.so", "librem2d_wide.so", "librem2d_fma.so")]


def main(argv):
    if len(argv) > 1 and argv[1] == "--scratch-map":
        lib = argv[2] if len(argv) > 2 else default_libraries()[0]
        for sym, rows in scratch_map(lib).items():
            print("# %s\n# bin instructions scratch_store scratch_load sgpr_spill_lane_moves lds global" % sym)
            for b, r in enumerate(rows):
                print("%2d %6d %4d %4d %4d %4d %4d" % tuple([b] + r))
        return 0
    libs = argv[1:] or default_libraries()
    report = {}
    for p in libs:
        report[os.path.relpath(p, ROOT)] = check_library(p)
    print(json.dumps(report, indent=1, sort_keys=True))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))

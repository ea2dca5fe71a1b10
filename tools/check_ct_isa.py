#!/usr/bin/env python3
"""Constant-time check on the gfx950 code object of libcheetah_hip.so (reference: the ct scan of
LookupTable::get_point, src/lookup.rs:81-102, and the `_vartime` naming convention, README.md:18).

The library's claim (include/cheetah_hip.h): in the constant-time kernels no branch and no
memory address depends on a scalar's window digit.  The compiler is free to turn a select into a branch, so the
claim is re-checked on the ISA of every build:

  1. the code object is taken out of the fat binary (llvm-objcopy + clang-offload-bundler) and disassembled
     (llvm-objdump -d);
  2. per kernel a forward taint analysis runs over the instruction stream (flow-sensitive per register, states merged
     at branch targets, iterated to a fixed point).  Sources are the digit reads: `ds_read_[iu]8` (digits recoded into
     LDS), `global_load_[su]byte` (the multi-scalar kernels' digits in global scratch) and, for the widened comb,
     which recodes on the fly, `v_and_b32 ..., 63, ...`.  Taint flows through every VALU / SALU result, carry masks,
     vcc, scc, EXEC (a VALU result written under a tainted EXEC is tainted), scratch spills and function results;
  3. violations: a conditional branch whose condition (vcc / scc / exec) is tainted, and a VMEM / LDS / scratch
     instruction whose address operand is tainted.  A read under a tainted EXEC at an untainted address -- the LDS
     lookups of the fixed-base kernels -- is allowed: which lanes take part in a wave-uniform access is not visible in
     the address stream.

The constant-time kernels must come out clean; the variable-time ones are the positive controls: the indexed-lookup
ladders (`k_multiply_ct<..., DIRECT>`) must show tainted addresses and the NAF kernels tainted branches, otherwise the
analysis itself has gone blind.  What it does NOT model: LDS contents (a digit written to LDS is only seen again
through the byte reads above) and timing differences of the instructions themselves (none of the integer VALU
instructions used has data-dependent timing).

    python tools/check_ct_isa.py [libcheetah_hip.so]     # prints a JSON summary; exit 1 on a violated expectation
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/llvm/bin"
DEFAULT_LIB = os.path.join(ROOT, "cheetah_amd", "lib", "libcheetah_hip.so")

CARRY_OUT = ("v_add_co_u32", "v_sub_co_u32", "v_subrev_co_u32", "v_addc_co_u32", "v_subb_co_u32", "v_subbrev_co_u32",
             "v_mad_u64_u32", "v_mad_i64_i32", "v_div_scale_f32", "v_div_scale_f64")
NO_SCC_WRITE = ("s_mov", "s_cmov", "s_cselect", "s_load", "s_getpc", "s_movk", "s_brev", "s_sext", "s_bfm", "s_mul_", "s_memtime",
                "s_getreg", "s_buffer_load", "s_pack")
SCC_READ = ("s_cselect", "s_cmov", "s_addc", "s_subb")


def disassemble(lib):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "code.co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
    return dict(zip(names, out.splitlines()))


def split_functions(text):
    funcs, cur, name, start = {}, None, None, 0
    for ln in text.splitlines():
        m = re.match(r"^([0-9a-f]+) <(\w+)>:", ln)
        if m:
            start, name = int(m.group(1), 16), m.group(2)
            cur = funcs.setdefault(name, {"start": start, "ins": []})
            continue
        if cur is None or not ln.startswith("\t"):
            continue
        m = re.match(r"^\t(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", ln)
        if not m:
            continue
        tgt = None
        mt = re.search(r"<\w+\+0x([0-9a-f]+)>\s*$", ln)
        if mt:
            tgt = start + int(mt.group(1), 16)
        elif re.search(r"<\w+>\s*$", ln) and m.group(1).startswith(("s_cbranch", "s_branch")):
            tgt = start
        cur["ins"].append((int(m.group(3), 16), m.group(1), m.group(2), tgt))
    return funcs


def regs(tok):
    """register names an operand token covers: 'v[2:3]' -> v2, v3; 'vcc' -> vcc; literals -> none"""
    tok = tok.strip()
    m = re.match(r"^(-?\|?)([vsa])(\d+)\|?$", tok)
    if m:
        return [m.group(2) + m.group(3)]
    m = re.match(r"^([vsa])\[(\d+):(\d+)\]$", tok)
    if m:
        return ["%s%d" % (m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    if tok in ("vcc", "vcc_lo", "vcc_hi"):
        return ["vcc"]
    if tok in ("exec", "exec_lo", "exec_hi"):
        return ["exec"]
    if tok in ("m0", "scc"):
        return [tok]
    return []


def operands(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch == "[":
            depth += 1
        if ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    # trailing modifiers ride on the last operand: 'v1 offset:16', 'off offset:4 sc0'
    clean = []
    for o in out:
        parts = o.split()
        clean.append(parts[0] if parts else "")
        clean_mod = " ".join(parts[1:])
        if clean_mod:
            clean.append("#" + clean_mod)
    return clean


WHY = None   # debugging: register -> the instruction that last tainted it


def analyse(fn, extra_sources=(), taint_args=0):
    """returns the list of violations [(addr, text, kind)] and counts"""
    ins = fn["ins"]
    index = {a: i for i, (a, _, _, _) in enumerate(ins)}
    incoming = {}                      # instruction index -> tainted set merged in from branches
    viol = {}
    sources = 0

    def is_source(mn, text):
        return bool(re.match(r"ds_read_[iu]8|global_load_[su]byte", mn)) or any(re.search(p, text) for p in extra_sources)

    # The ladder region: every kernel is a persistent loop over rounds of 256 (1 024) elements around the ladder.  Taint
    # must not travel from one round's RESULT into the next round (the result is output anyway, and whether it is the
    # identity is decided by predicated code whose EXEC reconvergence this analysis cannot resolve), so the back edge of
    # the outermost loop that holds the digit sources -- the round loop -- is not followed, and violations are looked for
    # from the first digit source to the end of the largest loops inside the round loop that hold one: the window loop(s)
    # with their lookups and additions, and the first window's lookup before them.
    src_idx = [i for i, (a, mn, ops, tg) in enumerate(ins) if is_source(mn, "%s %s" % (mn, ops))]
    loops = [(index[tg], i) for i, (a, mn, ops, tg) in enumerate(ins)
             if mn.startswith(("s_cbranch", "s_branch")) and tg is not None and tg in index and index[tg] <= i]
    cuts, region, inner = set(), (0, len(ins) - 1), []
    if src_idx and not taint_args:
        # round loops: hold every digit source AND a global store (the results leave there; a window loop stores nothing
        # to global memory).  A loop with several back edges appears once per edge; all of them are cut.
        def stores(lp):
            return any(ins[k][1].startswith(("global_store", "flat_store", "buffer_store")) for k in range(lp[0], lp[1] + 1))
        cuts = set(lp for lp in loops if all(lp[0] <= k <= lp[1] for k in src_idx) and stores(lp))
        heads = set(lp[0] for lp in cuts)
        cuts |= set(lp for lp in loops if lp[0] in heads)   # every back edge to a round loop's head
        inner = [lp for lp in loops if lp not in cuts and any(lp[0] <= k <= lp[1] for k in src_idx)]
        lo = min([src_idx[0]] + [lp[0] for lp in inner])
        hi = max([src_idx[-1]] + [lp[1] for lp in inner])
        region = (lo, hi)
    for _pass in range(12):
        changed = False
        T = set("v%d" % i for i in range(taint_args))
        sources = 0
        for i, (addr, mn, ops, tgt) in enumerate(ins):
            if i in incoming and not incoming[i] <= T:
                if WHY is not None:
                    for r in incoming[i] - T:
                        WHY[r] = "merged at 0x%x" % addr
                T |= incoming[i]
            op = [o for o in operands(ops)]
            mods = " ".join(o[1:] for o in op if o.startswith("#"))
            op = [o for o in op if not o.startswith("#")]

            def t(tokens):
                return any(r in T for tok in tokens for r in regs(tok))

            def setd(tokens, val):
                for tok in tokens:
                    for r in regs(tok):
                        (T.add if val else T.discard)(r)
                        if val and WHY is not None:
                            WHY[r] = "0x%x %s %s" % (addr, mn, ops)

            text = "%s %s" % (mn, ops)

            def note(k, a, txt, kind):
                if region[0] <= k <= region[1]:
                    viol[a] = (a, txt, kind)

            if mn.startswith(("s_cbranch", "s_branch")):
                cond = None
                if "vcc" in mn:
                    cond = "vcc"
                elif "scc" in mn:
                    cond = "scc"
                elif "exec" in mn:
                    cond = "exec"
                if cond and cond in T:
                    note(i, addr, text, "branch on " + cond)
                if tgt is not None and tgt in index and (index[tgt], i) not in cuts:
                    j = index[tgt]
                    if not T <= incoming.get(j, set()):
                        incoming[j] = incoming.get(j, set()) | T
                        if j <= i:
                            changed = True
                continue
            if mn.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_setprio", "s_sleep", "s_code_end", "s_set", "s_icache",
                              "s_dcache", "s_trap", "s_sendmsg", "s_inst_prefetch", "s_clause", "s_ttrace", "buffer_wbl2", "buffer_inv")):
                continue
            if mn.startswith("s_swappc"):
                # a call to fp6_mul_fn / fp6_square_fn / fp3_mul_fn: its result registers carry whatever came in
                if any(r.startswith("v") for r in T):
                    T |= set("v%d" % k for k in range(12))
                continue
            if mn.startswith("s_setpc"):
                if not taint_args:   # a long jump through a register: its target is not followed, so none may sit in the region
                    note(i, addr, text, "indirect branch")
                continue
            # ---- memory ----
            if mn.startswith(("global_load", "flat_load", "buffer_load")):
                a_t = t(op[1:3])
                if a_t:
                    note(i, addr, text, "address")
                src = is_source(mn, text)
                sources += src
                setd(op[0:1], src or a_t or "exec" in T)
                continue
            if mn.startswith(("global_store", "flat_store", "buffer_store", "global_atomic")):
                if t(op[0:1]) or t(op[2:3]):
                    note(i, addr, text, "address")
                continue
            if mn.startswith(("scratch_load", "scratch_store")):
                # spill slots are part of the state (strong updates, merged at branch targets like registers):
                # one name per dword of (vaddr, saddr, offset)
                load = mn.startswith("scratch_load")
                a_ops = op[1:3] if load else [op[0]] + op[2:3]
                if t(a_ops):
                    note(i, addr, text, "address")
                mo = re.search(r"offset:(\d+)", mods)
                off = int(mo.group(1)) if mo else 0
                mw = re.search(r"dwordx(\d)", mn)
                width = int(mw.group(1)) if mw else 1
                slots = ["M|%s|%d" % ("|".join(a_ops), off // 4 + k) for k in range(width)]
                if load:
                    setd(op[0:1], any(sl in T for sl in slots) or "exec" in T)
                else:
                    val = t(op[1:2]) or "exec" in T
                    for sl in slots:
                        (T.add if val else T.discard)(sl)
                continue
            if mn.startswith(("ds_read", "ds_load")):
                if t(op[1:2]):
                    note(i, addr, text, "address")
                src = is_source(mn, text)
                sources += src
                setd(op[0:1], src or t(op[1:2]) or "exec" in T)
                continue
            if mn.startswith(("ds_write", "ds_store")):
                if t(op[0:1]):
                    note(i, addr, text, "address")
                continue
            if mn.startswith(("ds_bpermute", "ds_permute", "ds_swizzle")):
                setd(op[0:1], t(op[1:]) or "exec" in T)
                continue
            # ---- scalar ALU ----
            if mn.startswith("s_"):
                base = mn
                if mn.startswith(("s_cmp", "s_bitcmp")):
                    (T.add if t(op) else T.discard)("scc")
                    continue
                if "saveexec" in mn:
                    old_exec = "exec" in T
                    new = t(op[1:]) or old_exec
                    setd(op[0:1], old_exec)
                    (T.add if new else T.discard)("exec")
                    (T.add if new else T.discard)("scc")
                    continue
                val = t(op[1:]) or (base.startswith(SCC_READ) and "scc" in T)
                if mn.startswith("s_or_b64") and op[0] == "exec" and op[1] == "exec":
                    val = t(op[2:3])   # reconvergence: exec is a subset of the saved mask it is or-ed back into
                setd(op[0:1], val)
                if not base.startswith(NO_SCC_WRITE):
                    (T.add if val else T.discard)("scc")
                continue
            # ---- vector ALU ----
            ex = "exec" in T
            if mn.startswith("v_cmpx"):
                val = t(op) or ex
                (T.add if val else T.discard)("exec")
                continue
            if mn.startswith(("v_readfirstlane", "v_readlane")):
                setd(op[0:1], t(op[1:]) or ex)
                continue
            nd = 2 if mn.startswith(CARRY_OUT) else 1
            val = t(op[nd:]) or ex
            if extra_sources and is_source(mn, text):
                val = True
                sources += 1
            if mn.startswith(("v_writelane",)):
                val = val or t(op[0:1])
            setd(op[0:nd], val)
        if not changed:
            break
    out = sorted(viol.values())
    return {"region": ["0x%x" % ins[region[0]][0], "0x%x" % ins[region[1]][0]], "round_loop_back_edges_cut": len(cuts), "window_loops": len(inner),
            "violations": [("0x%x" % a, txt, k) for a, txt, k in out],
            "branch": sum(1 for v in out if v[2].startswith("branch")), "address": sum(1 for v in out if v[2] == "address"),
            "digit_sources": sources, "instructions": len(ins)}


def lookup_hazards(fn):
    """The LDS lookups (lds_read_entry_where, kernels.hip.h) issue their ds_read_b128 from several inline-asm
    statements and wait ONCE, in a later statement: between a read and that wait the compiler believes the result
    registers hold their values (it does not see inside inline asm), so it could copy, spill or re-use them there.
    Checked on the ISA instead: from the first ds_read_b128 issued under `s_and_saveexec_b64` to the next
    `s_waitcnt lgkmcnt(0)`, only the lookup's own instructions (v_cmp_eq, s_and_saveexec, ds_read_b128, the EXEC
    restore, s_nop) may name a register a pending read writes."""
    own = ("ds_read_b128", "v_cmp_eq_u32", "s_and_saveexec_b64", "s_nop")
    pending, hazards, lookups, prev = set(), [], 0, ""
    for a, mn, ops, tgt in fn["ins"]:
        if mn.startswith("s_waitcnt") and "lgkmcnt(0)" in ops:
            if pending:
                lookups += 1
            pending = set()
        elif mn.startswith("ds_read_b128") and (pending or prev.startswith("s_and_saveexec_b64")):
            pending |= set(regs(operands(ops)[0]))
        elif pending and not mn.startswith(own) and not (mn.startswith("s_mov_b64") and ops.startswith("exec")):
            used = set(r for o in operands(ops) for r in regs(o))
            if used & pending:
                hazards.append(("0x%x" % a, "%s %s" % (mn, ops)))
        if not mn.startswith("s_nop"):
            prev = mn
    return lookups, hazards


# kernel name pattern (demangled) -> expectation
CT_KERNELS = [
    (r"k_multiply_ct<cheetah::\w+, \d, \d, false>", "ct", ()),
    (r"k_mul_by_generator<\d, (true|false)>", "ct", ()),
    (r"k_mul_by_generator_lookup<\d>", "ct", ()),
    (r"k_mul_by_generator_wide<(true|false)>", "ct", (r"^v_and_b32\S* v\d+, 63, v\d+",)),
    (r"k_multiply_many_ct<cheetah::\w+, \d, \d, false>", "ct", ()),
    (r"k_multiply_ct<cheetah::\w+, \d, \d, true>", "vt_address", ()),
    (r"k_multiply_many_ct<cheetah::\w+, \d, \d, true>", "vt_address", ()),
    (r"k_multiply_vartime<cheetah::\w+, \d, [012], false>", "vt_branch", ()),
]


def main(argv):
    lib = argv[1] if len(argv) > 1 else DEFAULT_LIB
    funcs = split_functions(disassemble(lib))
    names = demangle(list(funcs))
    report, ok = {}, True
    seen = {p: 0 for p, _, _ in CT_KERNELS}
    for mangled, fn in funcs.items():
        dn = names[mangled]
        for pat, kind, extra in CT_KERNELS:
            if re.search(pat, dn):
                seen[pat] += 1
                r = analyse(fn, extra)
                short = dn.split("(")[0].replace("void cheetah::", "")
                if kind == "ct":
                    good = not r["violations"] and r["digit_sources"] > 0 and r["window_loops"] > 0
                elif kind == "vt_address":
                    good = r["address"] > 0
                else:
                    good = r["branch"] > 0
                if "mul_by_generator" in dn:
                    nl, hz = lookup_hazards(fn)
                    good = good and nl > 0 and not hz
                    r["violations"] = r["violations"] + [(a, txt, "register of a pending LDS read") for a, txt in hz]
                ok = ok and good
                report[short] = {"expect": kind, "ok": good, "branch_violations": r["branch"], "address_violations": r["address"],
                                 "digit_sources": r["digit_sources"], "window_loops": r["window_loops"], "region": r["region"],
                                 "instructions": r["instructions"],
                                 "first": r["violations"][:3] if (kind == "ct" or not good) else []}
                break
    for name in ("_ZN7cheetah10fp6_mul_fnEDv6_mS0_", "_ZN7cheetah13fp6_square_fnEDv6_m"):
        # the field products the kernels call: no conditional branch, no memory access that depends on an argument,
        # every argument register taken as tainted
        if name in funcs:
            r = analyse(funcs[name], (), taint_args=24)
            good = r["branch"] == 0 and r["address"] == 0
            ok = ok and good
            report[names[name].split("(")[0].replace("cheetah::", "")] = {"expect": "ct (all arguments tainted)", "ok": good,
                                                                           "branch_violations": r["branch"], "address_violations": r["address"],
                                                                           "instructions": r["instructions"], "first": r["violations"][:3]}
        else:
            ok = False
            report[name] = {"ok": False, "missing": True}
    for pat, n in seen.items():
        if n == 0:
            ok = False
            report["no kernel matched " + pat] = {"ok": False}
    ver = subprocess.run([os.path.join(LLVM, "clang"), "--version"], capture_output=True, text=True).stdout.splitlines()[0]
    print(json.dumps({"ok": ok, "compiler": ver, "kernels": report}, indent=1))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv))

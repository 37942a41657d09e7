#!/usr/bin/env python3
"""Static check of the compiled tile kernels for a hazard the compiler missed once (letkf_tile2w.hip, UT = 1): a vector
or memory instruction reading (or overwriting) the result of a v_mfma fewer than 8 wait states after it, on ANY path -- the taken side of a branch
included (the compiler had padded only the fall-through side).  Usage: python tools/check_mfma_hazards.py file.hip [flags]
(compiles to assembly with hipcc -S --cuda-device-only and walks every kernel)."""
import re, subprocess, sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NEED = 8          # what the compiler itself pads v_mfma_f32_16x16x32_f16 (4 passes) -> vector reader to on straight-line code
# wait states an INTERVENING matrix instruction stands for: the matrix pipe of a SIMD takes its instructions in order and is busy 16
# cycles (4 passes) with a 16x16x32 half-precision one (MI355X_MICROARCH.md: ~16 cycles per MFMA back to back on one SIMD), so the one
# behind it cannot have started, let alone a vector instruction behind that, earlier than 4 issue slots after it
MFMA_WS = 4


def regs(tok):
    """Registers an operand token names, as (file, index) pairs: 'v' = vector registers, 'a' = accumulation registers (gfx90a+
    keeps both in one physical file, but the assembler's names -- and the hazards -- are per name space)."""
    tok = tok.strip().rstrip(",")
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return {(m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.fullmatch(r"([va])(\d+)", tok)
    return {(m.group(1), int(m.group(2)))} if m else set()


MATCHED = {"mfma": 0, "dst_regs": 0}       # (the test fails when a source holds matrix instructions whose operands were not parsed)


def operands(line):
    body = line.split(";")[0].strip()
    parts = body.split(None, 1)
    if len(parts) < 2:
        return parts[0] if parts else "", []
    return parts[0], [t for t in re.split(r",\s*", parts[1])]


def check(asm):
    lines = asm.splitlines()
    labels = {l.split(":")[0]: i for i, l in enumerate(lines) if re.match(r"^[.\w$]+:", l)}
    bad = []
    kernel = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\w+:", l):
            kernel = l.split(":")[0]
        op, ops = operands(l)
        if not op.startswith("v_mfma"):
            continue
        dst = regs(ops[0])
        MATCHED["mfma"] += 1
        MATCHED["dst_regs"] += len(dst)
        stack, seen = [(i + 1, 0)], set()
        while stack:
            j, ws = stack.pop()
            while j < len(lines) and ws < NEED:
                if (j, ws) in seen:
                    break
                seen.add((j, ws))
                o, a = operands(lines[j])
                if not o or o.endswith(":") or o.startswith(".") or o.startswith(";"):
                    j += 1
                    continue
                if o == "s_endpgm":
                    break
                if o.startswith("v_mfma"):
                    # a dependent MFMA (srcC or operand) is interlocked by other rules; an independent one does not read dst
                    pass
                elif o.startswith(("v_", "global_store", "ds_write", "buffer_store", "flat_store", "ds_bpermute", "ds_read", "global_load")):
                    # any read of the result, and any write over it (the matrix instruction's own write would land later)
                    if any(regs(t) & dst for t in a):
                        bad.append((kernel, i + 1, j + 1, ws, lines[i].strip(), lines[j].strip()))
                        break
                if o == "s_nop":
                    ws += int(a[0]) + 1
                elif o.startswith("v_mfma"):
                    ws += MFMA_WS        # the matrix pipe issues in order: a later matrix instruction starts >= 16 cycles after this one
                else:
                    ws += 1
                if o == "s_branch":
                    j = labels.get(a[0], len(lines))
                    continue
                if o.startswith("s_cbranch"):
                    stack.append((labels.get(a[0], len(lines)), ws))
                j += 1
    return bad


NEED_WAR = 3       # the compiler's own table (and what it pads its own code to): SrcC read of a 4-pass matrix instruction -> vector write over it


def check_war(asm):
    """A vector instruction (in practice: one inside an inline-asm statement, which the compiler's hazard recognizer does not look
    into) overwriting the accumulator INPUT of a matrix instruction fewer than NEED_WAR wait states after it."""
    lines = asm.splitlines()
    labels = {l.split(":")[0]: i for i, l in enumerate(lines) if re.match(r"^[.\w$]+:", l)}
    bad = []
    kernel = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\w+:", l):
            kernel = l.split(":")[0]
        op, ops = operands(l)
        if not op.startswith("v_mfma") or len(ops) < 4:
            continue
        src_c = regs(ops[3]) - regs(ops[0])          # (in-place accumulation: the instruction's own write is ordered)
        if not src_c:
            continue
        stack, seen = [(i + 1, 0)], set()
        while stack:
            j, ws = stack.pop()
            while j < len(lines) and ws < NEED_WAR:
                if (j, ws) in seen:
                    break
                seen.add((j, ws))
                o, a = operands(lines[j])
                if not o or o.endswith(":") or o.startswith(".") or o.startswith(";"):
                    j += 1
                    continue
                if o == "s_endpgm":
                    break
                if o.startswith("v_") and not o.startswith(("v_mfma", "v_cmp", "v_readlane", "v_readfirstlane")) and a and regs(a[0]) & src_c:
                    bad.append((kernel, i + 1, j + 1, ws, lines[i].strip(), lines[j].strip()))
                    break
                ws += int(a[0]) + 1 if o == "s_nop" else 1
                if o == "s_branch":
                    j = labels.get(a[0], len(lines))
                    continue
                if o.startswith("s_cbranch"):
                    stack.append((labels.get(a[0], len(lines)), ws))
                j += 1
    return bad


if __name__ == "__main__":
    src = sys.argv[1]
    # the flags the shipped object of this source is built with (torch-assimilate_amd/_build.py: SOURCE_FLAGS), then the caller's
    import importlib.util
    spec = importlib.util.spec_from_file_location("_mia_build", os.path.join(ROOT, "torch-assimilate_amd", "_build.py"))
    bld = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bld)
    flags = list(bld.SOURCE_FLAGS.get(os.path.basename(src), [])) + sys.argv[2:]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
               "-I" + os.path.join(ROOT, "torch-assimilate_amd", "csrc"), "--cuda-device-only", "-S", src, "-o", out] + flags
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            sys.exit("hipcc failed:\n" + res.stderr)
        text = open(out).read()
        bad = check(text)
        war = check_war(text)
    for b in bad[:40]:
        print("HAZARD kernel %s: mfma at line %d read at line %d after %d wait states\n   %s\n   %s" % b)
    for b in war[:40]:
        print("WAR HAZARD kernel %s: mfma at line %d, its SrcC overwritten at line %d after %d wait states\n   %s\n   %s" % b)
    print("%s: %d hazard(s), %d write-after-read hazard(s) on SrcC; %d matrix instructions, %d result registers parsed"
          % (os.path.basename(src), len(bad), len(war), MATCHED["mfma"], MATCHED["dst_regs"]))
    unparsed = "v_mfma" in text and (MATCHED["mfma"] == 0 or MATCHED["dst_regs"] < 4 * MATCHED["mfma"])
    if unparsed:
        print("CHECKER ERROR: the assembly holds matrix instructions whose result registers were not parsed")
    sys.exit(1 if bad or war or unparsed else 0)

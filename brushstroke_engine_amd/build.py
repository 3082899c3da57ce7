"""Ahead-of-time build of the gfx950 kernels into one C-ABI shared library (no JIT, no torch headers).

    python -m brushstroke_engine_amd.build            # build if stale
    python -m brushstroke_engine_amd.build --force

The reference JIT-compiles its plugins at first use (torch_utils/custom_ops.py:46-124); here the
library is built in-tree with ``hipcc --offload-arch=gfx950`` so that it travels with the source
snapshot and is visible as a loaded ``.so`` to whoever audits the process.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libneube_hip.so")
SOURCES = ["nb_ops.hip", "nb_modconv.hip", "nb_modconv_h3.hip", "nb_modconv_up2v.hip", "nb_modconv_small.hip", "nb_grad.hip", "nb_canvas.hip", "nb_encoder.hip", "nb_calib.hip"]
HEADERS = ["nb_common.h", "nb_h3_common.h", "nb_torgb.h", os.path.join("..", "..", "include", "neube_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall",
         "-Wno-unused-function", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt"]
# Per-file additions.  nb_ops.hip (ToRGB, bias_act, upfirdn2d, mapping ...) and nb_modconv.hip (exact-fp32 conv kernels): no SLP
# vectorisation -- it pairs their scalar arithmetic into swizzled packed fp32 instructions, which misbehave on this hardware
# (NB_NO_PACKED_F32 in csrc/nb_common.h has the story); these kernels have no use for packed arithmetic.  (The function attribute
# that removes packed fp32 per kernel costs the ToRGB kernel 384 bytes of scratch: helpers are no longer inlined into it.)
# nb_modconv_up2v.hip (round 4): their packed fp32 arithmetic is explicit (f32x4 / f32x2 vector types); SLP paired
# the two scalar `x * gain` of the prologue into a swizzled v_pk_mul_f32 (tests/test_abi.py scans for any such form).
FILE_FLAGS = {"nb_modconv.hip": ["-fno-slp-vectorize"], "nb_ops.hip": ["-fno-slp-vectorize"],
              "nb_modconv_up2v.hip": ["-fno-slp-vectorize"]}


STAMP = LIB + ".stamp"


def source_digest() -> str:
    """sha256 over the kernel sources, headers and compiler flags: what the library was (or would be) built from.
    Content, not mtimes -- a source snapshot copied to another machine keeps its digest."""
    import hashlib
    h = hashlib.sha256((" ".join(FLAGS) + repr(sorted(FILE_FLAGS.items()))).encode())
    for name in SOURCES + HEADERS:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()


def is_stale() -> bool:
    """True if there is no library or it was not built from the sources that are in the tree now."""
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return True
    with open(STAMP) as f:
        return f.read().strip() != source_digest()


def build(force: bool = False, verbose: bool = True) -> str:
    """Build if stale.  Safe under concurrent callers (torchrun ranks): an exclusive lock serialises them and the library
    is written to a per-process temporary before an atomic rename."""
    import fcntl
    if not force and not is_stale():
        return LIB
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not is_stale():                 # another process built it while this one waited
            return LIB
        digest = source_digest()
        tmp = f"{LIB}.{os.getpid()}.tmp"
        # one object per source file, compiled in parallel and cached by content (sources + headers + flags): editing one
        # kernel file recompiles that file only
        import hashlib
        from concurrent.futures import ThreadPoolExecutor
        objdir = os.path.join(CSRC, "build")
        os.makedirs(objdir, exist_ok=True)
        hdr = hashlib.sha256(" ".join(FLAGS).encode())
        for name in HEADERS:
            with open(os.path.join(CSRC, name), "rb") as f:
                hdr.update(f.read())
        cflags = [f for f in FLAGS if f != "-shared"]

        def compile_one(src):
            extra = FILE_FLAGS.get(src, [])
            with open(os.path.join(CSRC, src), "rb") as f:
                key = hashlib.sha256(hdr.digest() + " ".join(extra).encode() + f.read()).hexdigest()[:16]
            obj = os.path.join(objdir, f"{src}.{key}.o")
            if not os.path.exists(obj) or force:
                cmd = [HIPCC] + cflags + extra + ["-c", os.path.join(CSRC, src), "-o", f"{obj}.{os.getpid()}.tmp"]
                if verbose:
                    print("[build]", " ".join(cmd), flush=True)
                subprocess.check_call(cmd)
                os.replace(f"{obj}.{os.getpid()}.tmp", obj)
                for old in os.listdir(objdir):                     # drop older objects of this source
                    if old.startswith(src + ".") and old.endswith(".o") and os.path.join(objdir, old) != obj:
                        os.remove(os.path.join(objdir, old))
            return obj

        with ThreadPoolExecutor(max_workers=min(4, len(SOURCES))) as ex:
            objs = list(ex.map(compile_one, SOURCES))
        for old in os.listdir(objdir):                             # objects of sources that left the tree (tools/build_variant.sh links build/*.o)
            if old.endswith(".o") and old.split(".hip.")[0] + ".hip" not in SOURCES:
                os.remove(os.path.join(objdir, old))
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc"] + objs + ["-o", tmp]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        try:
            subprocess.check_call(cmd)
            os.replace(tmp, LIB)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
        with open(f"{STAMP}.{os.getpid()}.tmp", "w") as f:
            f.write(digest + "\n")
        os.replace(f"{STAMP}.{os.getpid()}.tmp", STAMP)
    return LIB


def kernel_resources(lib: str = LIB) -> dict:
    """Per-kernel register / scratch figures from the code-object metadata of the built library:
    {kernel name: dict(vgpr, sgpr_spill, vgpr_spill, scratch_bytes, lds_bytes)}.  The split-f16 kernels count their
    outstanding LDS-DMA operations (``s_waitcnt vmcnt(N)``), so compiler-made scratch traffic inside them is a
    defect to be caught at build time (tests/test_abi.py), not a performance footnote."""
    import re
    import shutil
    import tempfile
    llvm = os.path.join(os.path.dirname(os.path.dirname(HIPCC)), "lib", "llvm", "bin")
    tmp = tempfile.mkdtemp(prefix="nbres")
    try:
        work = os.path.join(tmp, "lib.so")
        shutil.copy(lib, work)
        subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", work], cwd=tmp, check=True, capture_output=True)
        out = {}
        for f in sorted(os.listdir(tmp)):
            if ARCH not in f:
                continue
            notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            for block in notes.split("- .agpr_count:")[1:]:
                def field(name):
                    m = re.search(r"\." + name + r":\s+(\S+)", block)
                    return m.group(1) if m else None
                out[field("name")] = dict(vgpr=int(field("vgpr_count")), sgpr_spill=int(field("sgpr_spill_count")), vgpr_spill=int(field("vgpr_spill_count")),
                                          scratch_bytes=int(field("private_segment_fixed_size")), lds_bytes=int(field("group_segment_fixed_size")))
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def disassembly(lib: str = LIB) -> str:
    """gfx950 disassembly of every code object in the built library (llvm-objdump; < 1 s)."""
    import shutil
    import tempfile
    llvm = os.path.join(os.path.dirname(os.path.dirname(HIPCC)), "lib", "llvm", "bin")
    tmp = tempfile.mkdtemp(prefix="nbdis")
    try:
        work = os.path.join(tmp, "lib.so")
        shutil.copy(lib, work)
        subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", work], cwd=tmp, check=True, capture_output=True)
        out = []
        for f in sorted(os.listdir(tmp)):
            if ARCH in f:
                out.append(subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", f"--mcpu={ARCH}", os.path.join(tmp, f)], check=True,
                                          capture_output=True, text=True).stdout)
        return "\n".join(out)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))


def mfma_loop_spill_traffic(lib: str = LIB) -> dict:
    """Scratch / SGPR-spill instructions INSIDE the matrix loops of every kernel of the built library, from its disassembly:
    {kernel name: [(mfma, scratch, v_readlane + v_writelane) per innermost loop that holds MFMA instructions]}.
    The persistent conv kernels of round 6 (a tile loop around the K loop) hold a few registers in scratch AROUND their K loops --
    spilled before a tile's loop, reloaded behind it, a dozen instructions per 30-50 us tile --; what must stay clean is the K loop
    itself, whose counted ``s_waitcnt vmcnt(N)`` would count a scratch access as one of its LDS-DMA pieces (tests/test_abi.py)."""
    import re
    import shutil
    import tempfile
    llvm = os.path.join(os.path.dirname(os.path.dirname(HIPCC)), "lib", "llvm", "bin")
    tmp = tempfile.mkdtemp(prefix="nbdis")
    out = {}
    try:
        work = os.path.join(tmp, "lib.so")
        shutil.copy(lib, work)
        subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", work], cwd=tmp, check=True, capture_output=True)
        for f in sorted(os.listdir(tmp)):
            if ARCH not in f:
                continue
            dis = subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            # functions: "<addr> <name>:" headers; instructions: "\tmnemonic operands // ADDR: encoding"
            for m in re.finditer(r"^[0-9a-f]+ <([^>]+)>:\n(.*?)(?=^[0-9a-f]+ <[^>]+>:\n|\Z)", dis, flags=re.M | re.S):
                name, body = m.group(1), m.group(2)
                if "v_mfma" not in body:
                    continue
                ins = []                                  # (address, text)
                for line in body.split("\n"):
                    mm = re.match(r"^\s+(\S.*?)\s*//\s*([0-9A-F]+):", line)
                    if mm:
                        ins.append((int(mm.group(2), 16), mm.group(1)))
                addr_index = {a: i for i, (a, _) in enumerate(ins)}
                spans = []
                for i, (a, t) in enumerate(ins):
                    bm = re.match(r"s_c?branch\S*\s+(\d+)", t)
                    if bm:                                # target = next pc + simm16 * 4
                        off = int(bm.group(1))
                        off = off - 65536 if off >= 32768 else off
                        tgt = a + 4 + 4 * off
                        if tgt <= a and tgt in addr_index:
                            spans.append((addr_index[tgt], i))
                inner = [sp for sp in set(spans) if not any(o != sp and o[0] >= sp[0] and o[1] <= sp[1] for o in spans)]
                rows = []
                for a0, a1 in sorted(inner):
                    seg = [t for _, t in ins[a0:a1 + 1]]
                    n_mf = sum(t.startswith("v_mfma") for t in seg)
                    # a K loop: matrix instructions and no output stores.  (The tile loop's closing jump lands behind the steady-state
                    # loops, so the stretch from there to the jump -- the last chunks' MFMAs, the whole epilogue -- looks innermost
                    # too; it runs once per tile and is told apart by its stores.)
                    if n_mf >= 8 and not any(t.startswith(("global_store", "buffer_store", "flat_store")) for t in seg):
                        rows.append((n_mf, sum(t.startswith("scratch_") for t in seg), sum(t.startswith(("v_readlane", "v_writelane")) for t in seg)))
                out[name] = rows
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)

#!/usr/bin/env python3
"""fix_pk_opsel.py -- works around a gfx950 (MI355X) hazard the compiler does not know, in a built HIP binary.

Measured (tools/pk_glitch.hip, profiles/r04_pk_glitch.txt): a packed float32 instruction -- v_pk_mul_f32,
v_pk_add_f32, v_pk_fma_f32 -- whose op_sel takes the HIGH register of the src1 pair for the LOW result while
src0 is taken straight (op_sel:[0,1], what the compiler emits for (a, b) * (d, c)) reads that operand as ZERO in
lanes 48-63 when the other wave of its SIMD issues an MFMA at the wrong cycle.  The same product with the
swizzle on src0 (op_sel:[1,0]) never does.  src0 and src1 of these three instructions commute, so every
occurrence is rewritten IN PLACE with its first two sources exchanged (registers, op_sel, op_sel_hi, neg_lo,
neg_hi): same arithmetic, same size, same schedule.

    python3 tools/fix_pk_opsel.py go-sdr_amd/libhzsdr_hip.so [more files ...]      (--check: report only, exit 1 if any)

The file may be a shared library / executable holding clang offload bundles (what hipcc links) or a bare
gfx950 code object.  csrc/Makefile and tools/Makefile run this behind every link; tests/test_capi_cpu.py runs
the check on the library.  VOP3P, gfx9 encoding (64 bits):
    dword 0: [7:0] vdst, [10:8] neg_hi, [13:11] op_sel, [14] op_sel_hi[2], [15] clamp, [22:16] op, [31:23] 0x1A7
    dword 1: [8:0] src0, [17:9] src1, [26:18] src2, [28:27] op_sel_hi[1:0], [31:29] neg_lo
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
PK = re.compile(r"^\s*(v_pk_(?:mul|add|fma)_f32)\s+(.*?)//\s*([0-9A-Fa-f]+):\s*([0-9A-Fa-f]{8})\s+([0-9A-Fa-f]{8})\s*$")


def code_objects(data):
    """(offset, size) of every gfx950 code object in `data`."""
    if data[:4] == b"\x7fELF" and struct.unpack_from("<H", data, 18)[0] == 224:  # EM_AMDGPU: a bare code object
        return [(0, len(data))]
    out, i = [], 0
    while True:
        i = data.find(MAGIC, i)
        if i < 0:
            return out
        (count,) = struct.unpack_from("<Q", data, i + 24)
        off = i + 32
        for _ in range(count):
            o, size, tl = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + tl].decode()
            off += tl
            if "amdgcn" in triple and size:
                if data[i + o:i + o + 4] != b"\x7fELF":
                    raise SystemExit("fix_pk_opsel: a bundle entry that is not an ELF (compressed bundle?): " + triple)
                out.append((i + o, size))
        i += len(MAGIC)


def text_sections(elf):
    """[(sh_addr, sh_offset, sh_size)] of the executable sections of an ELF64 image."""
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    secs = []
    for k in range(shnum):
        _, typ, flags, addr, off, size = struct.unpack_from("<IIQQQQ", elf, shoff + k * shentsize)
        if typ == 1 and flags & 4:  # SHT_PROGBITS, SHF_EXECINSTR
            secs.append((addr, off, size))
    return secs


def swap01(w0, w1):
    def swap_bits(w, a, b):
        x = ((w >> a) ^ (w >> b)) & 1
        return w ^ ((x << a) | (x << b))
    w0 = swap_bits(w0, 11, 12)  # op_sel
    w0 = swap_bits(w0, 8, 9)    # neg_hi
    s0, s1 = w1 & 0x1FF, (w1 >> 9) & 0x1FF
    w1 = (w1 & ~0x3FFFF) | s1 | (s0 << 9)
    w1 = swap_bits(w1, 27, 28)  # op_sel_hi
    w1 = swap_bits(w1, 29, 30)  # neg_lo
    return w0, w1


def hazardous(operands):
    m = re.search(r"op_sel:\[([01]),([01])", operands)
    return bool(m) and m.group(1) == "0" and m.group(2) == "1"


def process(path, check):
    data = bytearray(open(path, "rb").read())
    found = fixed = 0
    for base, size in code_objects(bytes(data)):
        elf = bytes(data[base:base + size])
        secs = text_sections(elf)
        with tempfile.NamedTemporaryFile(suffix=".elf", delete=False) as f:
            f.write(elf)
            tmp = f.name
        try:
            text = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", tmp], capture_output=True, text=True, check=True).stdout
        finally:
            os.unlink(tmp)
        for line in text.splitlines():
            m = PK.match(line)
            if not m or not hazardous(m.group(2)):
                continue
            found += 1
            if check:
                continue
            addr, w0, w1 = int(m.group(3), 16), int(m.group(4), 16), int(m.group(5), 16)
            at = next((off + addr - a for a, off, sz in secs if a <= addr < a + sz), None)
            if at is None or struct.unpack_from("<II", elf, at) != (w0, w1):
                raise SystemExit("fix_pk_opsel: %s: cannot place instruction at %#x" % (path, addr))
            if (w0 >> 23) != 0x1A7:
                raise SystemExit("fix_pk_opsel: %s: not a VOP3P encoding at %#x" % (path, addr))
            struct.pack_into("<II", data, base + at, *swap01(w0, w1))
            fixed += 1
    if fixed:
        tmp = path + ".pkfix"
        with open(tmp, "wb") as f:
            f.write(data)
        os.chmod(tmp, os.stat(path).st_mode)
        os.replace(tmp, path)
    return found, fixed


def main(argv):
    check = "--check" in argv
    files = [a for a in argv if not a.startswith("--")]
    if not files:
        raise SystemExit(__doc__)
    bad = 0
    for path in files:
        found, fixed = process(path, check)
        if check:
            print("%s: %d packed float32 instruction(s) with op_sel:[0,1]" % (path, found))
            bad += found
        else:
            left, _ = process(path, True)
            print("%s: %d rewritten, %d left" % (path, fixed, left))
            bad += left
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))

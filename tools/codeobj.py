"""Read the gfx950 code objects out of a built ``libgs_hip.so`` (no GPU needed).

Used by ``tests/test_isa_guard.py`` (the compiler-drift guard SURVEY.md §7 asks for) and from the
command line::

    python tools/codeobj.py [grayscott_amd/libgs_hip.so] [name filter]

For every kernel: the registers, spills, scratch and LDS bytes of its metadata note, the float mode
bits of its kernel descriptor (``compute_pgm_rsrc1``), and the disassembled instruction stream.
The host library carries one clang offload bundle per translation unit in ``.hip_fatbin``; each
bundle is split by hand (the bundler tool only sees the first), the gfx950 entries are ELF code
objects which ``llvm-readelf`` / ``llvm-objdump`` of the ROCm LLVM read.
"""
from __future__ import annotations

import os
import re
import struct
import subprocess
import sys
import tempfile
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import yaml

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "grayscott_amd", "libgs_hip.so")


@dataclass
class Kernel:
    symbol: str                      # mangled
    name: str                        # demangled, without the anonymous namespace and the argument list
    vgpr: int
    agpr: int
    sgpr: int
    vgpr_spill: int
    sgpr_spill: int
    scratch: int                     # .private_segment_fixed_size
    lds: int                         # .group_segment_fixed_size
    dynamic_stack: bool
    rsrc1: int = 0                   # compute_pgm_rsrc1 of the kernel descriptor
    insts: List[str] = field(default_factory=list)   # "mnemonic operands" per instruction (branch targets as labels)
    addrs: List[int] = field(default_factory=list)   # address of each instruction
    labels: Dict[str, int] = field(default_factory=dict)  # label -> address of the instruction it stands in front of

    # compute_pgm_rsrc1 fields (LLVM AMDGPUUsage, "compute_pgm_rsrc1 for GFX6-GFX12")
    @property
    def denorm_mode_32(self) -> int:
        return (self.rsrc1 >> 16) & 3

    @property
    def denorm_mode_16_64(self) -> int:
        return (self.rsrc1 >> 18) & 3

    @property
    def round_mode_32(self) -> int:
        return (self.rsrc1 >> 12) & 3

    @property
    def ieee_mode(self) -> int:
        return (self.rsrc1 >> 23) & 1

    @property
    def dx10_clamp(self) -> int:
        return (self.rsrc1 >> 21) & 1

    def count(self, pattern: str) -> int:
        rx = re.compile(pattern)
        return sum(1 for i in self.insts if rx.match(i))

    def matching(self, pattern: str) -> List[str]:
        rx = re.compile(pattern)
        return [i for i in self.insts if rx.match(i)]

    def loops(self) -> List[List[str]]:
        """The instruction ranges closed by a backward branch (a loop each, nested ones included), as instruction lists."""
        out = []
        for a, text in zip(self.addrs, self.insts):
            m = re.match(r"s_c?branch\w* (L\d+)", text)
            if m and m.group(1) in self.labels and self.labels[m.group(1)] <= a:
                lo = self.labels[m.group(1)]
                out.append([t for b, t in zip(self.addrs, self.insts) if lo <= b <= a])
        return out


def _tool(name: str) -> str:
    path = os.path.join(LLVM, name)
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not found: the ROCm LLVM tools are needed to read code objects")
    return path


def _elf_section(blob: bytes, wanted: str) -> bytes:
    """Contents of a named section of a little-endian ELF64 image."""
    if blob[:4] != b"\x7fELF" or blob[4] != 2:
        raise ValueError("not an ELF64 image")
    shoff, = struct.unpack_from("<Q", blob, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", blob, 0x3A)
    def sh(i):
        name, typ, flags, addr, off, size = struct.unpack_from("<IIQQQQ", blob, shoff + i * shentsize)
        return name, typ, addr, off, size
    _, _, _, stroff, strsize = sh(shstrndx)
    names = blob[stroff:stroff + strsize]
    for i in range(shnum):
        name, typ, addr, off, size = sh(i)
        end = names.index(b"\0", name)
        if names[name:end].decode() == wanted:
            return blob[off:off + size]
    raise KeyError(wanted)


def device_code_objects(lib: str = DEFAULT_LIB, arch: str = "gfx950") -> List[bytes]:
    """Every code object for ``arch`` in the library's ``.hip_fatbin``, one per translation unit."""
    with open(lib, "rb") as f:
        fat = _elf_section(f.read(), ".hip_fatbin")
    out = []
    pos = fat.find(MAGIC)
    while pos >= 0:
        count, = struct.unpack_from("<Q", fat, pos + len(MAGIC))
        o = pos + len(MAGIC) + 8
        for _ in range(count):
            off, size, tlen = struct.unpack_from("<QQQ", fat, o)
            o += 24
            triple = fat[o:o + tlen].decode()
            o += tlen
            if arch in triple and size:
                out.append(fat[pos + off:pos + off + size])
        pos = fat.find(MAGIC, pos + 1)
    return out


def _demangle(symbols: List[str]) -> List[str]:
    r = subprocess.run(["c++filt"], input="\n".join(symbols), capture_output=True, text=True, check=True)
    out = []
    for line in r.stdout.splitlines():
        line = line.replace("(anonymous namespace)::", "")
        line = re.sub(r"^void ", "", line)
        line = re.sub(r"\((?:GsStepArgs|GsWindowArgs|GsRunArgs|float|unsigned|int|char).*$", "", line)
        out.append(line.strip())
    return out


def _symbols(path: str) -> Dict[str, int]:
    """symbol name -> value, from llvm-readelf -s."""
    r = subprocess.run([_tool("llvm-readelf"), "-s", "-W", path], capture_output=True, text=True, check=True)
    out = {}
    for line in r.stdout.splitlines():
        parts = line.split()
        if len(parts) == 8 and parts[0].rstrip(":").isdigit():
            out[parts[7]] = int(parts[1], 16)
    return out


def kernels_of(code_object: bytes, disassemble: bool = True) -> List[Kernel]:
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "dev.co")
        with open(path, "wb") as f:
            f.write(code_object)
        notes = subprocess.run([_tool("llvm-readelf"), "--notes", path], capture_output=True, text=True,
                               check=True).stdout
        start = notes.index("---")
        end = notes.index("\n...", start) if "\n..." in notes[start:] else len(notes)
        meta = yaml.safe_load(notes[start + 3:end])
        ks = meta.get("amdhsa.kernels") or []
        names = _demangle([k[".name"] for k in ks])
        syms = _symbols(path)
        rodata_addr = None
        # .rodata holds the 64-byte kernel descriptors; its address = file offset in these objects' first segment,
        # but read it through the section table to be safe
        sec = subprocess.run([_tool("llvm-readelf"), "-S", "-W", path], capture_output=True, text=True,
                             check=True).stdout
        for line in sec.splitlines():
            m = re.search(r"\]\s+\.rodata\s+PROGBITS\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", line)
            if m:
                rodata_addr, rodata_off = int(m.group(1), 16), int(m.group(2), 16)
        out: List[Kernel] = []
        for k, name in zip(ks, names):
            kern = Kernel(symbol=k[".name"], name=name, vgpr=k[".vgpr_count"], agpr=k.get(".agpr_count", 0),
                          sgpr=k[".sgpr_count"], vgpr_spill=k.get(".vgpr_spill_count", 0),
                          sgpr_spill=k.get(".sgpr_spill_count", 0), scratch=k[".private_segment_fixed_size"],
                          lds=k[".group_segment_fixed_size"], dynamic_stack=bool(k.get(".uses_dynamic_stack", False)))
            kd = syms.get(k[".symbol"])
            if kd is not None and rodata_addr is not None:
                kern.rsrc1, = struct.unpack_from("<I", code_object, rodata_off + (kd - rodata_addr) + 48)
            out.append(kern)
        if disassemble:
            dis = subprocess.run([_tool("llvm-objdump"), "-d", "--symbolize-operands", path], capture_output=True,
                                 text=True, check=True).stdout
            by_symbol = {k.symbol: k for k in out}
            cur: Optional[Kernel] = None
            pending: List[str] = []
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    if re.fullmatch(r"L\d+", m.group(1)):
                        pending.append(m.group(1))      # a branch target inside the current kernel
                    else:
                        cur = by_symbol.get(m.group(1))
                        pending = []
                    continue
                if cur is None:
                    continue
                m = re.match(r"^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):", line)
                if m:
                    addr = int(m.group(2), 16)
                    for lab in pending:
                        cur.labels[lab] = addr
                    pending = []
                    cur.insts.append(re.sub(r"\s+", " ", m.group(1)))
                    cur.addrs.append(addr)
    return out


def kernels(lib: str = DEFAULT_LIB, disassemble: bool = True) -> List[Kernel]:
    out: List[Kernel] = []
    for co in device_code_objects(lib):
        out.extend(kernels_of(co, disassemble))
    return out


FLOAT_FMA = r"^v_(pk_)?(fma|fmac|fmamk|fmaak|mad|mac|madmk|madak)(_legacy)?_f(16|32|64)(?![0-9])|^v_fma_mix|^v_mad_mix|^v_dot"


def main(argv: List[str]) -> int:
    lib = argv[1] if len(argv) > 1 and os.path.exists(argv[1]) else DEFAULT_LIB
    filt = argv[-1] if len(argv) > 1 and argv[-1] != lib else ""
    for k in kernels(lib):
        if filt and not re.search(filt, k.name):
            continue
        print(f"{k.name}: vgpr {k.vgpr} sgpr {k.sgpr} spills v{k.vgpr_spill}/s{k.sgpr_spill} scratch {k.scratch} "
              f"lds {k.lds} denorm32 {k.denorm_mode_32} ieee {k.ieee_mode} insts {len(k.insts)} "
              f"fma {k.count(FLOAT_FMA)} readlane/writelane {k.count(r'^v_(readlane|writelane)_b32')} "
              f"scratch_ops {k.count(r'^scratch_')}")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))

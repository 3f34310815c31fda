"""Register / scratch reports of compiled kernel translation units.

hipcc's `-Rpass-analysis=kernel-resource-usage` remarks (one `.usage` file per TU, written by
csrc/Makefile and codegen/plugin.py) give registers, spills and the size of the private segment.
A non-zero ScratchSize does not mean that the kernel touches scratch memory: at the register limit
LLVM reserves an emergency frame (68 bytes per lane on the kernels seen here) that no instruction
accesses.  What must never ship is a kernel that really spills to scratch memory while it also
spills SGPRs (the combination that miscomputed under hipcc 7.2), so for
kernels that report a private segment the device code is disassembled and its scratch
instructions are counted.
"""
from __future__ import annotations

import glob
import os
import re
import subprocess
import tempfile
from typing import Dict, List

LLVM_BIN = os.environ.get("BRBD_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
_SCRATCH = re.compile(r"^\s*(scratch_(?:load|store)\w*|buffer_(?:load|store)_\w+\s+\S+,\s*off,\s*s\[0:3\])", re.M)


def parse_usage(path: str) -> List[Dict[str, str]]:
    rows, cur = [], None
    for line in open(path):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    return [r for r in rows if "k_" in r["name"]]


def scratch_instructions(obj: str) -> Dict[str, int]:
    """kernel symbol -> number of instructions that access scratch memory, from the disassembly of
    the gfx950 code object bundled in a host object file."""
    out: Dict[str, int] = {}
    with tempfile.TemporaryDirectory(prefix="brbd_dis_") as d:
        # (the bundles are written next to the input file: work on a link inside the scratch directory)
        os.symlink(os.path.abspath(obj), os.path.join(d, "tu.o"))
        subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "--offloading", "tu.o"], cwd=d,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        cos = [f for f in glob.glob(os.path.join(d, "*")) if "amdgcn" in os.path.basename(f)]
        for co in cos:
            text = subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", co], stdout=subprocess.PIPE,
                                  stderr=subprocess.DEVNULL, check=True).stdout.decode(errors="replace")
            cur = None
            for line in text.splitlines():
                m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
                if m:
                    cur = m.group(1)
                    out.setdefault(cur, 0)
                elif cur is not None and _SCRATCH.match(line.split("//")[0]):
                    out[cur] += 1
    return out


def kernel_reports(gen_dir: str, obj_suffix: str = ".o") -> List[Dict[str, str]]:
    """One dict per compiled kernel of a build directory: the remark fields plus `ScratchInsts`
    (scratch-memory instructions in its code; disassembled only when ScratchSize > 0)."""
    rows = []
    for u in sorted(glob.glob(os.path.join(gen_dir, "*.usage"))):
        ks = parse_usage(u)
        need = [k for k in ks if int(k.get("ScratchSize", 0)) > 0]
        counts = {}
        if need:
            obj = u[:-len(".usage")] + obj_suffix
            if os.path.exists(obj):
                counts = scratch_instructions(obj)
        for k in ks:
            if int(k.get("ScratchSize", 0)) > 0:
                k["ScratchInsts"] = str(counts.get(k["name"], -1))  # -1: could not be disassembled
            else:
                k["ScratchInsts"] = "0"
            rows.append(k)
    return rows


def refused(k: Dict[str, str]) -> bool:
    """scratch memory really used (or not verifiable) together with more than 8 spilled SGPRs"""
    return int(k.get("ScratchInsts", 0)) != 0 and int(k.get("SGPRs Spill", 0)) > 8

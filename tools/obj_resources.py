#!/usr/bin/env python3
"""Registers / spills / scratch and the number of full memory drains (s_waitcnt vmcnt(0)) of every kernel in a BUILT object
(coarse3d_amd/csrc/*.o) -- no recompilation: the gfx950 code object is taken out of the object's .hip_fatbin section.
usage: tools/obj_resources.py coarse3d_amd/csrc/wgrad_tr.o [name filter]
A kernel that spills reloads from scratch with the reload as the YOUNGEST memory operation, so each reload in a pipelined
loop is an s_waitcnt vmcnt(0): it drains every global load in flight (round 5: the fused weight-gradient instances)."""
import os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
obj = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as tmp:
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "k.co")
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, capture_output=True)
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
meta = {}
for blk in notes.split("  - .agpr_count:")[1:]:
    g = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1))      # noqa: E731
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    meta[name] = dict(vgpr=g("vgpr_count"), sgpr=g("sgpr_count"), spill=g("vgpr_spill_count"), scratch=g("private_segment_fixed_size"))
drain, scr, mfma = {}, {}, {}
cur = None
for line in dis.splitlines():
    m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
    if m:
        cur = m.group(1)
        drain[cur] = scr[cur] = mfma[cur] = 0
        continue
    if cur is None:
        continue
    t = line.strip()
    if "s_waitcnt vmcnt(0)" in t:
        drain[cur] += 1
    elif "scratch_" in t:
        scr[cur] += 1
    elif "v_mfma" in t:
        mfma[cur] += 1
names = subprocess.run(["c++filt"], input="\n".join(meta), capture_output=True, text=True).stdout.splitlines()
for mangled, nice in zip(meta, names):
    nice = nice.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    if flt and flt not in nice:
        continue
    r = meta[mangled]
    print(f"{nice[:70]:70s} vgpr {r['vgpr']:4d} spill {r['spill']:4d} scratch {r['scratch']:4d} B  scratch-ops {scr.get(mangled, 0):4d}  "
          f"vmcnt(0) {drain.get(mangled, 0):4d}  mfma {mfma.get(mangled, 0):4d}")

"""Register / scratch / LDS metadata of the kernels in a hipcc object (gfx950 code object notes).
    python profiles/kmeta.py quandary_amd/csrc/build/qd_q32.o [substring]"""
import os, re, subprocess, sys, tempfile
L = "/opt/rocm/lib/llvm/bin"
obj = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
t = tempfile.mkdtemp()
subprocess.check_call([f"{L}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, f"{t}/fat.bin"])
for trg in ("hipv4-amdgcn-amd-amdhsa--gfx950", "hip-amdgcn-amd-amdhsa--gfx950"):
    r = subprocess.run([f"{L}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={t}/fat.bin", f"--targets={trg}", f"--output={t}/dev.co"], capture_output=True)
    if r.returncode == 0 and os.path.getsize(f"{t}/dev.co") > 0:
        break
txt = subprocess.check_output([f"{L}/llvm-readelf", "--notes", f"{t}/dev.co"], text=True)
cur = {}
for line in txt.splitlines():
    m = re.match(r"\s+[- ]\s*\.(\w+):\s+(.*)$", line) or re.match(r"\s+\.(\w+):\s+(.*)$", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2).strip()
    if k in ("name", "vgpr_count", "vgpr_spill_count", "sgpr_count", "sgpr_spill_count", "private_segment_fixed_size", "agpr_count", "symbol"):
        cur[k] = v
    if k == "symbol":
        pass
    if k == "wavefront_size":
        name = subprocess.check_output(["c++filt", cur.get("name", "?").strip("'")], text=True).strip()
        if pat in name and "k_" in name:
            print(f"{name[:110]:110s} vgpr {cur.get('vgpr_count')} agpr {cur.get('agpr_count')} spill {cur.get('vgpr_spill_count')} scratch {cur.get('private_segment_fixed_size')} sgpr {cur.get('sgpr_count')}")
        cur = {}

"""Registers, scratch and LDS of every kernel in the built objects, from the code objects' own metadata (llvm-readelf --notes
of the gfx950 code object inside each .o) -- what the occupancy arithmetic in DESIGN.md rests on.  rocprofv3's VGPR_Count
column prints HALF the allocated vector registers on this stack (80 for a kernel whose metadata says 155 -> 160 allocated).
usage: python tools/kernel_resources.py [> profiles/r04/kernel_resources.txt]"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = "/opt/rocm/lib/llvm/bin/"
rows = []
with tempfile.TemporaryDirectory() as tmp:
    for obj in sorted(glob.glob(os.path.join(ROOT, "ark_ec_vrfs_amd", "csrc", "*.o"))):
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "k.co")
        if subprocess.run([L + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj], capture_output=True).returncode:
            continue
        r = subprocess.run([L + "clang-offload-bundler", "--type=o", "--unbundle", "--input=" + fat,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True)
        if r.returncode or not os.path.exists(co):
            continue
        notes = subprocess.run([L + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            g = lambda key: (re.search(r"\." + key + r":\s+(\S+)", blk) or [None, "?"])[1]
            name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(.*$", "", name.replace("(anonymous namespace)::", "")).replace("vrf::f_bls381fr::", "").replace("vrf::f_25519::", "f1::").replace(
                "vrf::f_bn254fr::", "f2::").replace("vrf::", "").replace("(anonymous namespace)::", "").replace("void ", "")
            v = int(g("vgpr_count"))
            a = int(re.match(r"\s*(\d+)", blk).group(1))
            tot = v                          # .vgpr_count is the unified total (architectural + accumulation registers)
            alloc = (tot + 7) // 8 * 8
            waves = min(8, 512 // alloc) if alloc else 8
            rows.append((os.path.basename(obj), name, v, a, int(g("private_segment_fixed_size")), int(g("group_segment_fixed_size")),
                         int(g("max_flat_workgroup_size")), waves))
        os.remove(co)
print("%-22s %-78s %5s %5s %8s %8s %6s %s" % ("object", "kernel", "regs", "agpr", "scratch", "lds", "maxwg", "waves/SIMD by registers"))
for r in rows:
    print("%-22s %-78s %5d %5d %8d %8d %6d %d" % (r[0], r[1][:78], *r[2:]))

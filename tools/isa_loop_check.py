"""Where a kernel's scratch accesses, barriers and `s_waitcnt vmcnt(0)` sit in its gfx950 assembly (developer tool).
usage: isa_loop_check.py FILE.hip KERNEL_SUBSTRING  — compiles adhoc-queries-pointclouds_amd/csrc/FILE.hip to assembly and prints,
for the first kernel whose mangled name contains the substring, every scratch_*, s_barrier, global_load/store and vmcnt wait with
its line offset, so that one can see what lies inside the hot loop."""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "adhoc-queries-pointclouds_amd", "csrc", sys.argv[1])
out = "/tmp/isa_check.s"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-I" + os.path.join(root, "include"),
                "-I" + os.path.dirname(src), "-S", "--cuda-device-only", "-o", out, src], check=True, capture_output=True)
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and sys.argv[2] in l)
end = start
while not lines[end].startswith(".Lfunc_end"):
    end += 1
print(lines[start][:100], end - start, "lines")
only = set(sys.argv[3].split(",")) if len(sys.argv) > 3 else None
for n in range(start, end):
    l = lines[n].strip()
    kind = "scratch" if l.startswith("scratch_") else "barrier" if l.startswith("s_barrier") else "vmem" if l.startswith(("global_", "buffer_", "flat_")) else \
        "wait0" if "vmcnt(0)" in l else "wait" if "vmcnt(" in l else None
    if kind and (only is None or kind in only):
        print(f"{n - start:6d} {kind:8s} {l.split(';')[0][:80]}")

"""The loops of a kernel in its gfx950 assembly, with what each holds (developer tool).
usage: isa_loops.py FILE.hip KERNEL_SUBSTRING — compiles adhoc-queries-pointclouds_amd/csrc/FILE.hip to assembly and prints, for
the first kernel whose mangled name contains the substring, every loop (a backward branch and its target) with its length,
vector instructions, v_readlane/v_writelane (scalar registers parked in vector lanes: each fetch is a vector instruction),
LDS and memory instructions, scratch accesses and barriers.  A loop that holds another one counts the inner one's too."""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "adhoc-queries-pointclouds_amd", "csrc", sys.argv[1])
out = "/tmp/isa_loops.s"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-I" + os.path.join(root, "include"),
                "-I" + os.path.dirname(src), "-S", "--cuda-device-only", "-o", out, src], check=True, capture_output=True)
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and sys.argv[2] in l)
end = start
while not lines[end].startswith(".Lfunc_end"):
    end += 1
body = lines[start:end]
print(body[0][:110], len(body), "lines")
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
loops = set()
for i, l in enumerate(body):
    m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.add((labels[m.group(1)], i))
def count(a, b, pred):
    return sum(1 for l in body[a:b] if pred(l.strip()))
print("%6s %6s %6s %6s %8s %9s %5s %6s %8s %8s" % ("from", "to", "lines", "valu", "readlane", "writelane", "lds", "vmem", "scratch", "barrier"))
for a, b in sorted(loops):
    print("%6d %6d %6d %6d %8d %9d %5d %6d %8d %8d" % (a, b, b - a, count(a, b, lambda l: l.startswith("v_")), count(a, b, lambda l: l.startswith("v_readlane")),
          count(a, b, lambda l: l.startswith("v_writelane")), count(a, b, lambda l: l.startswith("ds_")), count(a, b, lambda l: l.startswith(("global_", "buffer_", "flat_"))),
          count(a, b, lambda l: l.startswith("scratch_")), count(a, b, lambda l: l.startswith("s_barrier"))))

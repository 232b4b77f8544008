"""Dev tool: static instruction histogram of one kernel by (inlined) source function.
usage: python tools/static_profile.py [kernel-substring]   (compiles with -gline-tables-only into /tmp)"""
import bisect, collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
kern = sys.argv[1] if len(sys.argv) > 1 else "k_step_groupILi16ELb1"
tmp = "/tmp/meshenv_static"; os.makedirs(tmp, exist_ok=True)
src = os.path.join(ROOT, "reinforcementlearning4meshgeneration_amd", "csrc")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
                       "-fno-fast-math", "--offload-arch=gfx950", "-mllvm", "-disable-machine-licm", "-mllvm",
                       "-amdgpu-atomic-optimizer-strategy=None", "-gline-tables-only", "--cuda-device-only", "-c", "-o", tmp + "/dev.o",
                       src + "/meshenv_hip.hip"], stderr=subprocess.DEVNULL)
subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + tmp + "/dev.o",
                       "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + tmp + "/dev_gfx950.o"])
dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "-l", tmp + "/dev_gfx950.o"], capture_output=True, text=True).stdout.splitlines()
start = next(i for i, l in enumerate(dis) if kern in l and l.endswith(">:"))
end = next((i for i in range(start + 1, len(dis)) if re.match(r"^[0-9a-f]{16} <", dis[i])), len(dis))
cur, hist, cls_hist, op_hist = None, collections.Counter(), collections.Counter(), collections.Counter()
def klass(op):
    if op.startswith("s_"):
        return "smem" if op.startswith(("s_load", "s_buffer")) else ("branch" if op.startswith(("s_cbranch", "s_branch")) else ("wait/nop" if op.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_barrier")) else "salu"))
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    return "vmem"
for l in dis[start:end]:
    m = re.match(r"^; (/.*):(\d+)$", l.strip())
    if m:
        cur = (os.path.basename(m.group(1)), int(m.group(2)))
    else:
        m2 = re.match(r"^\s+([a-z_0-9]+) ", l)
        if m2 and cur:
            hist[cur] += 1
            cls_hist[(cur, klass(m2.group(1)))] += 1
            if klass(m2.group(1)) == "salu": op_hist[m2.group(1)] += 1
def funcs_of(path):
    out = []
    for i, l in enumerate(open(path).read().splitlines(), 1):
        if re.match(r"^(__device__|__global__|template|k_step)", l):
            names = re.findall(r"\b([a-zA-Z_0-9]+)\(", l)
            if names: out.append((i, names[0]))
    return out
tables = {f: funcs_of(os.path.join(src, f)) for f in os.listdir(src) if f.endswith(".h")}
agg, agg_cls = collections.Counter(), collections.Counter()
def fn_of(f, ln):
    if f in tables and tables[f]:
        k = bisect.bisect_right([x[0] for x in tables[f]], ln) - 1
        return f.replace("meshenv_", "") + ":" + (tables[f][k][1] if k >= 0 else "?")
    return f + ":" + str(ln)
for (f, ln), c in hist.items():
    agg[fn_of(f, ln)] += c
for ((f, ln), kl), c in cls_hist.items():
    agg_cls[(fn_of(f, ln), kl)] += c
tot = sum(agg.values())
print(dis[start].split("<")[1][:60], "static instructions:", tot)
classes = ["valu", "salu", "branch", "wait/nop", "smem", "lds", "vmem"]
print("       total   %   " + " ".join(f"{k:>8s}" for k in classes))
for k, c in agg.most_common(36):
    print(f"{c:6d} {100 * c / tot:5.1f}%  " + " ".join(f"{agg_cls[(k, kl)]:8d}" for kl in classes) + "  " + k)
print("all   ", " " * 7, " ".join(f"{sum(v for (kk, kl2), v in agg_cls.items() if kl2 == kl):8d}" for kl in classes))
print("scalar-ALU opcodes:", ", ".join(f"{k} {v}" for k, v in op_hist.most_common(24)))

"""Register / scratch / LDS use of every kernel of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage):
    python tools/kernel_resources.py attention [conv_gemm ...]      # prints name, VGPR, AGPR, SGPR, scratch bytes, occupancy, LDS"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for unit in sys.argv[1:]:
    src = os.path.join(ROOT, "stablediffusioneo_amd", "csrc", unit + ".hip")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                        "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    cur = {}
    for line in r.stderr.splitlines():
        m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r":\d+:\d+: remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
        elif ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
            if k.strip().startswith("LDS Size"):
                name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
                name = re.sub(r"\(.*$", "", name).replace("void ", "")
                print(f"{name[:70]:70s} vgpr {cur.get('VGPRs','?'):>4s} agpr {cur.get('AGPRs','?'):>4s} sgpr {cur.get('SGPRs','?'):>4s} "
                      f"scratch {cur.get('ScratchSize [bytes/lane]','?'):>5s} occ {cur.get('Occupancy [waves/SIMD]','?'):>2s} lds {cur.get('LDS Size [bytes/block]','?')}")

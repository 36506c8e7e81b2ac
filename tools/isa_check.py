"""ISA invariants of the persistent ping-pong kernels, checked on the compiler's output (no GPU needed):

  1. no `s_waitcnt vmcnt(0)` directly in front of the hand-written fragment reads (ds_read_b128 asm blocks) of a K loop.  hipcc's wait-count
     pass puts one there when it believes a compiler-visible VMEM operation may still be pending at the loop header (an epilogue load whose
     result is only read inside a conditional region, a visible store, ...); it drains the LDS-DMA prefetch ring in front of EVERY K-tile.
     Round 4 found it in the EPI_F32X kernels (present since round 3) and in a rewritten EPI_RESID epilogue (-3 % on fc2 / out-proj).
  2. no scratch memory beyond a few spilled registers (an accumulator array that a non-inlined lambda takes by reference lives in scratch).

usage: python tools/isa_check.py [file.hip ...]      (default: the product library's persistent kernels)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "law_of_vision_representation_in_mllms_amd", "csrc")
FILES = sys.argv[1:] or ["gemm_bf16_v5.hip", "gemm_bf16_v2.hip", "ascore.hip"]
bad = 0
for f in FILES:
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "--cuda-device-only", "-S",
               os.path.join(CSRC, f), "-o", tmp.name]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr[-2000:])
        lines = open(tmp.name).read().splitlines()
    kernel, wait_at = None, None
    hits, scratch = {}, {}
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            kernel = m.group(1)
        t = ln.strip()
        if t.startswith("s_waitcnt vmcnt(0)"):
            wait_at = i
        elif t.startswith("ds_read_b128"):
            if wait_at is not None and i - wait_at <= 2 and "ASMSTART" in lines[i - 1]:
                hits[kernel] = hits.get(kernel, 0) + 1
            wait_at = None
        m = re.search(r"\.private_segment_fixed_size:\s*(\d+)", ln) or re.search(r"; ScratchSize: (\d+)", ln)
        if m and kernel:
            scratch[kernel] = max(scratch.get(kernel, 0), int(m.group(1)))
    for k, n in hits.items():
        print(f"{f}: {k}: {n} K-loop header(s) behind s_waitcnt vmcnt(0)")
        bad += 1
    for k, n in scratch.items():
        if n > 128:
            print(f"{f}: {k}: {n} bytes of scratch per lane")
            bad += 1
    print(f"{f}: {len(scratch)} kernels checked")
sys.exit(1 if bad else 0)

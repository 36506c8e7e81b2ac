"""Where a tile round of the persistent 256x256 GEMM goes, and at which clock (diagnostic build: build.build_variant_lib('tiletiming', ['-DV5_TILE_TIMING'], only=['gemm_bf16_v5.hip']), VISREP_LIB=.../libvisrep_hip_tiletiming.so;
the K loop itself is the production one):
shader-clock stamps of block 0's wave 0 (group 0) and wave 4 (group 1) at the tile boundaries - the K loop of a tile (first fragment read to the
counted wait behind its last MFMA segment) and the rest (closing barrier(s) + epilogue + accumulator reset), averaged over the block's tiles."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from law_of_vision_representation_in_mllms_amd import _lib, engine
dev = "cuda:0"
lib = _lib.load()
buf = torch.zeros(16 + 4 * 1024 + 32, dtype=torch.int64, device=dev)      # 16 sums of block 0 + (start tick, end tick, cycles, tiles) per block
lib.visrep_debug_gemm_timing_buffer(_lib.ptr(buf))
M = 147456
for (N, K, epi, tag) in ((4096, 1024, "act", "fc1"), (2048, 1024, "bias", "Q|K"), (1024, 1024, "resid", "out"), (1024, 4096, "resid", "fc2"), (1024, 1024, "vt", "V^T")):
    a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    o = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    def run():
        if epi == "act": engine.gemm(a, w, bias, _lib.EPI_ACT, act="quick_gelu", out=o)
        elif epi == "bias": engine.gemm(a, w, bias, _lib.EPI_BIAS, out=o)
        elif epi == "resid": engine.gemm(a, w, bias, _lib.EPI_RESID, resid=o, out=o)
        else: engine.linear_vt(a, w, bias)
    for bal in (0, 1):                                      # equal shares of the tile list, then the XCD-weighted split (visrep_set_xcd_balance)
        lib.visrep_set_xcd_balance(bal)
        for _ in range(100 if bal else 3):                  # one launch in 16 is a measurement: let the estimate settle
          buf.zero_(); run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        buf.zero_(); run(); torch.cuda.synchronize()
        t = buf.cpu().tolist()
        print(f"{tag:4s} xcd balance {'on ' if bal else 'off'} {_lib.xcd_balance()}", flush=True)
        blk = torch.tensor(t[16:16 + 4096]).reshape(-1, 4)
        blk = blk[blk[:, 3] > 0].double()
        if len(blk):
            t0 = blk[:, 0].min()
            st, en = (blk[:, 0] - t0) * 0.01, (blk[:, 1] - t0) * 0.01            # us since the first block's first tile
            dur = en - st
            xcd = torch.arange(len(blk)) % 8
            print(f"{tag:4s} {len(blk)} blocks: tile loops start {st.min():.1f} .. {st.max():.1f} us, end {en.min():.1f} .. {en.max():.1f} us (launch {ms * 1e3:.1f} us); "
                  f"loop duration min / median / max {dur.min():.1f} / {dur.median():.1f} / {dur.max():.1f} us; "
                  f"per XCD (block % 8) mean end " + " ".join(f"{en[xcd == x].mean():.1f}" for x in range(8))
                  + "; clock per XCD " + " ".join(f"{(blk[xcd == x, 2] / ((blk[xcd == x, 1] - blk[xcd == x, 0]) * 10)).mean():.3f}" for x in range(8)), flush=True)
        for g in range(2):
            kb = t[16 + 4096 + g * 16: 16 + 4096 + g * 16 + 16]
            print(f"{tag:4s} group {g}: cycles per K-tile iteration (closing barrier included; the last one carries the epilogue), by position in the output tile: "
                  + ", ".join(f"{nm} {kb[i] / max(kb[8 + i], 1):.0f}" for i, nm in enumerate(("first", "second", "third", "middle", "last"))) + "  (matrix pipe alone: 2048)", flush=True)
        for g in range(2):
            loop, epi_c, n, bar = t[g * 8], t[g * 8 + 1], max(t[g * 8 + 2], 1), t[g * 8 + 3]
            tot = loop + epi_c
            print(f"{tag:4s} group {g}: {n} tiles  K loop {loop / n:8.0f} cyc ({loop / n / (K // 64):6.0f} per K-tile)  boundary {epi_c / n:8.0f} cyc = {100 * epi_c / max(tot, 1):4.1f} % of the block's time"
                  f" [epilogue body {(epi_c - bar) / n:6.0f} cyc, waiting at its barrier(s) {bar / n:6.0f} cyc];  launch {ms:.4f} ms;"
                  f"  shader clock {t[g * 8 + 4] / max(t[g * 8 + 5], 1) / 10.0:.3f} GHz (s_memtime / s_memrealtime over the block's tiles);"
                  f"  matrix pipe busy {100.0 * n * (K // 64) * 2048 / max(t[g * 8 + 4], 1):.1f} % of those cycles (2 waves x 64 MFMAs x 16 cycles per K-tile and SIMD)", flush=True)

"""Board telemetry of the GPU a process computes on, read from amdgpu's sysfs files (no root needed): socket power, shader clock, busy
percentage at ~100 Hz from hwmon, and - from the binary `gpu_metrics` table - the firmware's energy accumulator and the accumulated
residencies of its throttlers (PPT = package power tracking, i.e. the board power limit; PROCHOT; socket / VR / HBM thermal).

Measurement plumbing only (bench.py's `roofline.power`, tools/power_sample.py -> profiles/round6_power.md): nothing on the scoring path imports
it.  Why it exists: profiles/round5_gemm.md argued from in-kernel clock stamps that the headline GEMMs sit on the board's power limit; the
round-5 review asked for the board's own word on it.

gpu_metrics layout (Linux amdgpu `kgd_pp_interface.h`, struct gpu_metrics_v1_x; little endian):
  header: u16 structure_size, u8 format_revision, u8 content_revision
  v1.4 / v1.5: 6 x u16 (temperature_hotspot, temperature_mem, temperature_vrsoc, curr_socket_power [W], average_gfx_activity, average_umc_activity),
        u16 vcn_activity[4], u16 jpeg_activity[32 | 8 for 1.4], u64 energy_accumulator [15.259 uJ], u64 system_clock_counter [ns], u32 throttle_status, ...
  v1.6: the same six u16, then u64 energy_accumulator, u64 system_clock_counter, u32 accumulation_counter, u32 prochot_residency_acc,
        u32 ppt_residency_acc, u32 socket_thm_residency_acc, u32 vr_thm_residency_acc, u32 hbm_thm_residency_acc, u32 gfxclk_lock_status, ...
  v1.7 / v1.8 (MI355X boxes of round 6 report 1.8): as 1.6 with u64 mem_max_bandwidth in front of energy_accumulator (checked on the box: the
        word at offset 16 is constant, the one at offset 24 advances by socket power x time / 15.259 uJ)
The decode is cross-checked against hwmon in `selfcheck()` (socket power and hotspot temperature must agree) before anything trusts it.
"""
from __future__ import annotations

import glob
import os
import struct
import threading
import time
from typing import Dict, List, Optional

ENERGY_UNIT_J = 15.259e-6            # gpu_metrics energy_accumulator unit


def _read(path: str) -> Optional[str]:
    try:
        with open(path) as fh:
            return fh.read().strip()
    except OSError:
        return None


def _pci_of_torch_device(index: int) -> Optional[str]:
    try:
        import torch
        p = torch.cuda.get_device_properties(index)
        dom, bus, dev = getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", None), getattr(p, "pci_device_id", 0)
        if bus is None:
            return None
        return f"{dom:04x}:{bus:02x}:{dev:02x}.0"
    except Exception:
        return None


def find_card(index: int = 0) -> Optional[str]:
    """/sys/class/drm/cardN/device of HIP device `index`: matched by PCI address; with one hwmon-bearing card, that card."""
    cards = [c for c in sorted(glob.glob("/sys/class/drm/card*/device")) if glob.glob(os.path.join(c, "hwmon", "hwmon*"))]
    if not cards:
        return None
    want = _pci_of_torch_device(index)
    if want:
        for c in cards:
            if os.path.basename(os.path.realpath(c)).lower() == want.lower():
                return c
    return cards[0] if len(cards) == 1 else None


def decode_gpu_metrics(blob: bytes) -> Dict[str, object]:
    if len(blob) < 64:
        return {"error": f"gpu_metrics: {len(blob)} bytes"}
    size, fmt, rev = struct.unpack_from("<HBB", blob, 0)
    out: Dict[str, object] = {"version": f"{fmt}.{rev}", "structure_size": size}
    if fmt != 1:
        out["error"] = "unknown format revision"
        return out
    hot, mem, vr, power, gfx_act, umc_act = struct.unpack_from("<6H", blob, 4)
    out.update(temperature_hotspot_c=hot, temperature_mem_c=mem, socket_power_w=power, gfx_activity_pct=gfx_act, umc_activity_pct=umc_act)
    if rev >= 6:
        e, clk, acc, prochot, ppt, sthm, vrthm, hbmthm = struct.unpack_from("<QQ6I", blob, 24 if rev >= 7 else 16)
        out.update(energy_acc=e, system_clock_ns=clk, accumulation_counter=acc, prochot_residency_acc=prochot, ppt_residency_acc=ppt,
                   socket_thm_residency_acc=sthm, vr_thm_residency_acc=vrthm, hbm_thm_residency_acc=hbmthm)
    elif rev in (4, 5):
        off = 16 + 2 * 4 + 2 * (32 if rev == 5 else 8)
        off = (off + 7) // 8 * 8
        e, clk, thr = struct.unpack_from("<QQI", blob, off)
        out.update(energy_acc=e, system_clock_ns=clk, throttle_status=thr)
    else:
        out["error"] = "content revision not decoded"
    return out


class Board:
    """The sysfs files of one GPU."""

    def __init__(self, index: int = 0):
        self.card = find_card(index)
        self.hwmon = None
        if self.card:
            h = sorted(glob.glob(os.path.join(self.card, "hwmon", "hwmon*")))
            self.hwmon = h[0] if h else None

    @property
    def ok(self) -> bool:
        return bool(self.hwmon)

    def power_w(self) -> Optional[float]:
        v = _read(os.path.join(self.hwmon, "power1_input")) or _read(os.path.join(self.hwmon, "power1_average"))
        return float(v) * 1e-6 if v else None

    def power_cap_w(self) -> Optional[float]:
        v = _read(os.path.join(self.hwmon, "power1_cap"))
        return float(v) * 1e-6 if v else None

    def sclk_mhz(self) -> Optional[float]:
        v = _read(os.path.join(self.hwmon, "freq1_input"))
        return float(v) * 1e-6 if v else None

    def busy_pct(self) -> Optional[float]:
        v = _read(os.path.join(self.card, "gpu_busy_percent"))
        return float(v) if v else None

    def temp_c(self) -> Optional[float]:
        v = _read(os.path.join(self.hwmon, "temp2_input"))
        return float(v) * 1e-3 if v else None

    def metrics(self) -> Dict[str, object]:
        try:
            with open(os.path.join(self.card, "gpu_metrics"), "rb") as fh:
                return decode_gpu_metrics(fh.read())
        except OSError as e:
            return {"error": str(e)}

    def selfcheck(self) -> Dict[str, object]:
        """Does the gpu_metrics decode agree with hwmon?  (socket power within 15 % or 30 W, hotspot temperature within 3 C)"""
        m, p, t = self.metrics(), self.power_w(), self.temp_c()
        ok = "error" not in m and p is not None and abs(float(m["socket_power_w"]) - p) <= max(30.0, 0.15 * p) and (t is None or abs(float(m["temperature_hotspot_c"]) - t) <= 3.0)
        return {"decode_trusted": bool(ok), "gpu_metrics_version": m.get("version"), "hwmon_power_w": p, "metrics_power_w": m.get("socket_power_w"),
                "hwmon_hotspot_c": t, "metrics_hotspot_c": m.get("temperature_hotspot_c")}


class Sampler:
    """Background thread: (t, power W, sclk MHz, busy %) as fast as sysfs answers (each read is a firmware query: ~100-300 Hz in all), and the
    gpu_metrics accumulators at start() and stop() - the firmware's own energy / throttler-residency integrals over the window."""

    def __init__(self, board: Optional[Board] = None, index: int = 0, period_s: float = 0.005):
        self.board = board or Board(index)
        self.period = period_s
        self.rows: List[tuple] = []
        self._stop = threading.Event()
        self._th: Optional[threading.Thread] = None
        self.m0: Dict[str, object] = {}
        self.m1: Dict[str, object] = {}
        self.t0 = self.t1 = 0.0

    def _run(self):
        b = self.board
        while not self._stop.is_set():
            t = time.perf_counter()
            self.rows.append((t, b.power_w(), b.sclk_mhz(), b.busy_pct()))
            dt = self.period - (time.perf_counter() - t)
            if dt > 0:
                self._stop.wait(dt)

    def start(self):
        if not self.board.ok:
            return self
        self.rows = []
        self._stop.clear()
        self.m0 = self.board.metrics()
        self.t0 = time.perf_counter()
        self._th = threading.Thread(target=self._run, daemon=True)
        self._th.start()
        return self

    def stop(self) -> Dict[str, object]:
        if not self.board.ok or self._th is None:
            return {"available": False}
        self._stop.set()
        self._th.join()
        self.t1 = time.perf_counter()
        self.m1 = self.board.metrics()
        return self.summary()

    def summary(self, skip_s: float = 0.0) -> Dict[str, object]:
        rows = [r for r in self.rows if r[0] - self.t0 >= skip_s and r[1] is not None]
        if not rows:
            return {"available": False}

        def stat(i):
            v = sorted(r[i] for r in rows if r[i] is not None)
            if not v:
                return None
            return {"mean": round(sum(v) / len(v), 1), "p50": round(v[len(v) // 2], 1), "p95": round(v[min(len(v) - 1, int(0.95 * len(v)))], 1), "max": round(v[-1], 1), "min": round(v[0], 1)}
        cap = self.board.power_cap_w()
        out: Dict[str, object] = {"available": True, "samples": len(rows), "window_s": round(self.t1 - self.t0, 3), "hz": round(len(self.rows) / max(self.t1 - self.t0, 1e-9), 1),
                                  "power_w": stat(1), "power_cap_w": cap, "sclk_mhz": stat(2), "busy_pct": stat(3)}
        if cap and out["power_w"]:
            out["frac_of_cap_samples_ge_95pct"] = round(sum(1 for r in rows if r[1] >= 0.95 * cap) / len(rows), 3)
        m0, m1 = self.m0, self.m1
        if "error" not in m0 and "error" not in m1 and "energy_acc" in m0 and "energy_acc" in m1:
            dt_ns = int(m1["system_clock_ns"]) - int(m0["system_clock_ns"])
            de = (int(m1["energy_acc"]) - int(m0["energy_acc"])) * ENERGY_UNIT_J
            acc = {"gpu_metrics_version": m1.get("version"), "energy_j": round(de, 1)}
            if dt_ns > 0:
                acc["firmware_window_s"] = round(dt_ns * 1e-9, 3)
                acc["mean_power_w_from_energy"] = round(de / (dt_ns * 1e-9), 1)
            if "accumulation_counter" in m0:
                n = (int(m1["accumulation_counter"]) - int(m0["accumulation_counter"])) & 0xFFFFFFFF      # u32 counters: differences modulo 2^32
                acc["accumulation_cycles"] = n
                for k in ("ppt", "prochot", "socket_thm", "vr_thm", "hbm_thm"):
                    dv = (int(m1[f"{k}_residency_acc"]) - int(m0[f"{k}_residency_acc"])) & 0xFFFFFFFF
                    acc[f"{k}_residency"] = round(dv / n, 4) if n > 0 else None      # share of the firmware's accumulation cycles the throttler was active
            elif "throttle_status" in m1:
                acc["throttle_status_end"] = hex(int(m1["throttle_status"]))
            out["firmware"] = acc
        return out


def measure(fn, seconds: float = 2.0, index: int = 0, settle_s: float = 0.3) -> Dict[str, object]:
    """Run fn() repeatedly for about `seconds` (after `settle_s` of untimed repeats so that the clock has settled) with the sampler on."""
    import torch
    t_end = time.perf_counter() + settle_s
    while time.perf_counter() < t_end:
        fn()
    torch.cuda.synchronize()
    s = Sampler(index=index).start()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            fn()
        n += 8
        torch.cuda.synchronize()
    out = s.stop()
    out["launch_groups"] = n
    return out

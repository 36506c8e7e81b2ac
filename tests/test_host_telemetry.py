"""Board-telemetry plumbing (law_of_vision_representation_in_mllms_amd/telemetry.py: bench.py's roofline.power, tools/power_sample.py) on CPU: the gpu_metrics
decode against hand-packed tables of the layouts it claims to know, the sampler's summary arithmetic on a fake board, and its behaviour
on a box without an amdgpu card (this container): unavailable, never an exception."""
import struct
import time

from law_of_vision_representation_in_mllms_amd import telemetry as T


def _blob(rev, power=1387, hot=71, energy=123456789, clk=5_000_000_000, acc=1000, ppt=900):
    head = struct.pack("<HBB", 256, 1, rev) + struct.pack("<6H", hot, 60, 55, power, 99, 40)
    if rev >= 7:
        body = struct.pack("<Q", 8_000_000) + struct.pack("<QQ6I", energy, clk, acc, 3, ppt, 0, 0, 0)            # mem_max_bandwidth in front (v1.7 / v1.8)
    else:
        body = struct.pack("<QQ6I", energy, clk, acc, 3, ppt, 0, 0, 0)
    return (head + body).ljust(256, b"\0")


def test_gpu_metrics_decode_v16_and_v18():
    for rev in (6, 8):
        m = T.decode_gpu_metrics(_blob(rev))
        assert m["version"] == f"1.{rev}" and m["socket_power_w"] == 1387 and m["temperature_hotspot_c"] == 71
        assert m["energy_acc"] == 123456789 and m["system_clock_ns"] == 5_000_000_000
        assert m["accumulation_counter"] == 1000 and m["ppt_residency_acc"] == 900 and m["prochot_residency_acc"] == 3
    assert "error" in T.decode_gpu_metrics(b"\0" * 10)
    assert "error" in T.decode_gpu_metrics(struct.pack("<HBB", 64, 2, 0).ljust(64, b"\0"))


class FakeBoard:
    ok, card = True, "fake"

    def __init__(self):
        self.n = 0

    def power_w(self):
        self.n += 1
        return 1390.0 if self.n % 2 else 1300.0

    def power_cap_w(self):
        return 1400.0

    def sclk_mhz(self):
        return 1950.0

    def busy_pct(self):
        return 100.0

    def metrics(self):
        k = self.n
        return T.decode_gpu_metrics(_blob(8, energy=1000 + 100 * k, clk=10 ** 9 * (1 + k), acc=1000 + 10 * k, ppt=900 + 9 * k))


def test_sampler_summary_on_a_fake_board():
    s = T.Sampler(FakeBoard(), period_s=0.001).start()
    time.sleep(0.05)
    out = s.stop()
    assert out["available"] and out["samples"] >= 5 and out["power_cap_w"] == 1400.0
    assert 1300.0 <= out["power_w"]["mean"] <= 1390.0 and out["power_w"]["max"] == 1390.0 and out["sclk_mhz"]["mean"] == 1950.0
    assert 0.3 <= out["frac_of_cap_samples_ge_95pct"] <= 0.7              # every other sample is above 0.95 x cap
    fw = out["firmware"]
    assert abs(fw["ppt_residency"] - 0.9) < 1e-6 and fw["energy_j"] >= 0 and fw["firmware_window_s"] > 0


def test_no_card_is_unavailable_not_an_error():
    b = T.Board(0)
    if b.ok:                                                               # a box with an amdgpu card: the decode must agree with hwmon
        assert b.selfcheck()["gpu_metrics_version"] is not None
        return
    s = T.Sampler(b).start()
    assert s.stop() == {"available": False}

"""Board power / shader-clock telemetry for the benchmarks: a sampling thread over the amdgpu hwmon files of one device.

Measurement infrastructure only (bench.py, tools/): nothing on the generator's path imports it.  The reference has no
counterpart (its only timer is forger/util/timer.py:11-32).  The files are plain sysfs text, readable by an ordinary
user on the GPU boxes: ``/sys/class/drm/card*/device/hwmon/hwmon*/power1_input`` (micro-watts, falls back to
``power1_average``) and ``freq1_input`` (Hz, label sclk).  No GPU call is made from here.
"""
from __future__ import annotations

import glob
import os
import threading
import time
from typing import Dict, List, Optional


def _read_int(path: str) -> Optional[int]:
    try:
        with open(path) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return None


def find_hwmon(pci_bus_id: Optional[str] = None) -> Optional[str]:
    """hwmon directory of the amdgpu device with that PCI address ("0000:05:00.0"); without an address, of the first device that
    has one.  An address that matches no device gives None (no telemetry) rather than another GPU's sensors."""
    first = None
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
        dev = os.path.join(card, "device")
        mons = sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*")))
        if not mons:
            continue
        first = first or mons[0]
        if pci_bus_id:
            try:
                if os.path.basename(os.path.realpath(dev)).lower() == pci_bus_id.lower():
                    return mons[0]
            except OSError:
                pass
    return None if pci_bus_id else first


def pci_bus_id_of(device_index: int) -> Optional[str]:
    """PCI address of a torch device, from its properties (no kernel launch)."""
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        return "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
    except Exception:                                   # noqa: BLE001
        return None


class PowerClockSampler:
    """``with PowerClockSampler(hwmon) as s: ...; s.mark("f8"); run; s.unmark()`` -- samples (time, watts, MHz) every
    ``period`` seconds in a daemon thread; ``summary(name)`` gives the means over the marked window."""

    def __init__(self, hwmon: Optional[str], period: float = 0.01):
        self.hwmon = hwmon
        self.period = period
        self.samples: List[tuple] = []
        self.windows: Dict[str, List[float]] = {}
        self._stop = threading.Event()
        self._thread: Optional[threading.Thread] = None
        self.power_file = None
        if hwmon:
            for name in ("power1_input", "power1_average"):
                if _read_int(os.path.join(hwmon, name)) is not None:
                    self.power_file = os.path.join(hwmon, name)
                    break
            self.freq_file = os.path.join(hwmon, "freq1_input") if _read_int(os.path.join(hwmon, "freq1_input")) is not None else None
            self.cap_w = (_read_int(os.path.join(hwmon, "power1_cap")) or 0) / 1e6 or None
        else:
            self.freq_file, self.cap_w = None, None

    @property
    def available(self) -> bool:
        return bool(self.power_file or self.freq_file)

    def _run(self):
        while not self._stop.is_set():
            t = time.perf_counter()
            w = _read_int(self.power_file) if self.power_file else None
            f = _read_int(self.freq_file) if self.freq_file else None
            self.samples.append((t, None if w is None else w / 1e6, None if f is None else f / 1e6))
            self._stop.wait(self.period)

    def __enter__(self):
        if self.available:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=1.0)
        return False

    def mark(self, name: str):
        self.windows[name] = [time.perf_counter(), float("inf")]

    def unmark(self, name: str):
        """Close (or move on) the end of window ``name``; its start stays where ``mark`` put it."""
        if name in self.windows:
            self.windows[name][1] = time.perf_counter()

    MIN_SAMPLES = 20          # means over fewer ticks are noise (a 36 ms window sees 3-4): summary() then reports the count only

    def summary(self, name: str) -> dict:
        if not self.available:
            return {"power_w_mean": None, "sclk_mhz_mean": None, "samples": 0, "source": "no readable amdgpu hwmon files on this box"}
        t0, t1 = self.windows.get(name, [0.0, 0.0])
        sel = [s for s in list(self.samples) if t0 <= s[0] <= t1]
        if len(sel) < self.MIN_SAMPLES:
            return {"samples": len(sel), "window_ms": round((t1 - t0) * 1e3, 2),
                    "source": f"fewer than {self.MIN_SAMPLES} samples in the window: no power / clock means reported"}
        ws = [s[1] for s in sel if s[1] is not None]
        fs = [s[2] for s in sel if s[2] is not None]
        return {"power_w_mean": round(sum(ws) / len(ws), 1) if ws else None, "power_w_max": round(max(ws), 1) if ws else None,
                "sclk_mhz_mean": round(sum(fs) / len(fs), 1) if fs else None, "sclk_mhz_min": round(min(fs), 1) if fs else None,
                "samples": len(sel), "window_ms": round((t1 - t0) * 1e3, 2), "power_cap_w": self.cap_w,
                "source": f"{self.power_file or '-'} / {self.freq_file or '-'} sampled every {self.period * 1e3:.0f} ms by a side thread "
                          f"(sysfs sclk reads up to ~10 % above the in-kernel clock: MI355X_MICROARCH.md DVFS item 6)"}

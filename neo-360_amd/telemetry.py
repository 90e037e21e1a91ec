"""Socket power and shader clock of one GPU, sampled in-process while a benchmark's timed region runs.

The split-fp16 evaluators run at the socket's power limit: the clock the part sustains is part of
every throughput number, so `bench.py` puts both into its JSON line (VERDICT r3 task 4) instead of leaving them to
builder-side `rocm-smi` logs.  Source: librocm_smi64.so through ctypes (the calls behind `rocm-smi --showpower
--showclocks`), polled from a daemon thread; nothing here touches the HIP stream being measured.
"""
import ctypes
import os
import threading
import time

_RSMI_CLK_TYPE_SYS = 0
_MAX_FREQ = 33


class _Frequencies(ctypes.Structure):
    _fields_ = [("has_deep_sleep", ctypes.c_bool), ("num_supported", ctypes.c_int32), ("current", ctypes.c_uint32),
                ("frequency", ctypes.c_uint64 * _MAX_FREQ)]


_lib = None
_lib_err = None


def _load():
    global _lib, _lib_err
    if _lib is not None or _lib_err is not None:
        return _lib
    for path in (os.environ.get("NEO360_RSMI_LIB"), "/opt/rocm/lib/librocm_smi64.so", "librocm_smi64.so"):
        if not path:
            continue
        try:
            lib = ctypes.CDLL(path)
            if lib.rsmi_init(ctypes.c_uint64(0)) != 0:
                raise OSError("rsmi_init failed")
            _lib = lib
            return _lib
        except OSError as e:           # no library / no driver (the build container): telemetry is simply absent
            _lib_err = str(e)
    return None


def _rsmi_index(lib, torch_index):
    """ROCm-SMI's device index for torch's device `torch_index`: matched on the PCI bus id (HIP_VISIBLE_DEVICES may
    renumber), falling back to the same ordinal."""
    try:
        import torch
        bus = getattr(torch.cuda.get_device_properties(torch_index), "pci_bus_id", None)
        n = ctypes.c_uint32(0)
        if bus is not None and lib.rsmi_num_monitor_devices(ctypes.byref(n)) == 0:
            for i in range(n.value):
                bdf = ctypes.c_uint64(0)
                if lib.rsmi_dev_pci_id_get(ctypes.c_uint32(i), ctypes.byref(bdf)) == 0 and ((bdf.value >> 8) & 0xFF) == int(bus):
                    return i
    except Exception:
        pass
    return torch_index


def read_once(torch_index=0):
    """(power_w, sclk_mhz, power_limit_w) right now; any of them None when the query is unsupported."""
    lib = _load()
    if lib is None:
        return None, None, None
    dv = ctypes.c_uint32(_rsmi_index(lib, torch_index))
    return _power(lib, dv), _sclk(lib, dv), _cap(lib, dv)


def _power(lib, dv):
    p, kind = ctypes.c_int64(0), ctypes.c_int(0)
    try:
        if lib.rsmi_dev_power_get(dv, ctypes.byref(p), ctypes.byref(kind)) == 0:
            return p.value / 1e6
    except AttributeError:
        pass
    p = ctypes.c_uint64(0)
    try:
        if lib.rsmi_dev_current_socket_power_get(dv, ctypes.byref(p)) == 0:
            return p.value / 1e6
    except AttributeError:
        pass
    return None


def _sclk(lib, dv):
    f = _Frequencies()
    if lib.rsmi_dev_gpu_clk_freq_get(dv, ctypes.c_int(_RSMI_CLK_TYPE_SYS), ctypes.byref(f)) == 0 and f.current < _MAX_FREQ:
        return f.frequency[f.current] / 1e6
    return None


def _cap(lib, dv):
    c = ctypes.c_uint64(0)
    if lib.rsmi_dev_power_cap_get(dv, ctypes.c_uint32(0), ctypes.byref(c)) == 0:
        return c.value / 1e6
    return None


def _energy_uj(lib, dv):
    """The socket's energy accumulator in microjoules (rsmi_dev_energy_count_get: counter x resolution), or None."""
    e, res, ts = ctypes.c_uint64(0), ctypes.c_float(0.0), ctypes.c_uint64(0)
    try:
        if lib.rsmi_dev_energy_count_get(dv, ctypes.byref(e), ctypes.byref(res), ctypes.byref(ts)) == 0 and res.value > 0:
            return e.value * float(res.value)
    except AttributeError:
        pass
    return None


class Sampler:
    """with Sampler(device_index) as s: ... timed region ...; s.summary() -> means over the samples taken inside."""

    def __init__(self, torch_index=0, period_s=0.05):
        self.period = period_s
        self.power, self.sclk = [], []
        self.cap = None
        self.energy_j = None            # socket energy accumulator over the region (round 6): joules, not power x time
        self._e0 = self._t0 = self.wall_s = None
        self._stop = threading.Event()
        self._thread = None
        self._lib = _load()
        self._dv = ctypes.c_uint32(_rsmi_index(self._lib, torch_index)) if self._lib is not None else None

    def __enter__(self):
        if self._lib is not None:
            self.cap = _cap(self._lib, self._dv)
            self._e0, self._t0 = _energy_uj(self._lib, self._dv), time.perf_counter()
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()
        return self

    def _run(self):
        while not self._stop.is_set():
            p, c = _power(self._lib, self._dv), _sclk(self._lib, self._dv)
            if p is not None:
                self.power.append(p)
            if c is not None:
                self.sclk.append(c)
            self._stop.wait(self.period)

    def __exit__(self, *exc):
        if self._lib is not None and self._e0 is not None:
            e1 = _energy_uj(self._lib, self._dv)
            self.wall_s = time.perf_counter() - self._t0
            if e1 is not None and e1 >= self._e0:
                self.energy_j = (e1 - self._e0) * 1e-6
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=2.0)
        return False

    def summary(self):
        """{} when librocm_smi64 / the driver is not there (nothing is invented)."""
        if self._lib is None:
            return {"telemetry": "unavailable (%s)" % (_lib_err or "librocm_smi64 not loaded")}
        mean = lambda xs: (sum(xs) / len(xs)) if xs else None
        return {"sclk_mhz_mean": mean(self.sclk), "sclk_mhz_min": min(self.sclk) if self.sclk else None,
                "power_w_mean": mean(self.power), "power_w_max": max(self.power) if self.power else None,
                "power_limit_w": self.cap, "telemetry_samples": len(self.power),
                "energy_j": self.energy_j, "energy_window_s": self.wall_s,
                "power_w_from_energy": (self.energy_j / self.wall_s) if self.energy_j and self.wall_s else None,
                "telemetry": "librocm_smi64 (rsmi_dev_power_get, rsmi_dev_gpu_clk_freq_get SYS), polled every %d ms from a "
                             "thread of the bench process during the timed steps" % int(self.period * 1e3)}

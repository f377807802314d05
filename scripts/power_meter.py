"""Package power / energy beside a running workload, read in-process from ROCm SMI (librocm_smi64.so) over ctypes -- measurement plumbing for
bench.py (`power` object of the JSON line) and the A/B scripts; not part of the product path.

  with PowerMeter(device=0, period_s=0.02) as pm:     # a sampling thread: socket power (W) and the PLL sclk every 20 ms
      workload(); torch.cuda.synchronize()
  pm.summary()  -> {"watts": {median, p10, p90, max}, "cap_watts", "energy_j" (the device's energy accumulator over the window),
                    "mean_watts_from_energy", "sclk_mhz_reported": {...}, "samples", "window_s"}

Every field is best-effort: a call the SMI library does not support on the box is reported as null, never raised (bench.py must not lose
its headline line to a telemetry failure)."""
import ctypes as C
import threading
import time


class _Freqs(C.Structure):
    _fields_ = [("has_deep_sleep", C.c_bool), ("num_supported", C.c_uint32), ("current", C.c_uint32), ("frequency", C.c_uint64 * 33)]


class PowerMeter:
    def __init__(self, device: int = 0, period_s: float = 0.02):
        self.dev, self.period = C.c_uint32(device), period_s
        self.lib = C.CDLL("librocm_smi64.so")
        st = self.lib.rsmi_init(C.c_uint64(0))
        if st != 0:
            raise RuntimeError(f"rsmi_init -> {st}")
        self.watts, self.sclk, self.t = [], [], []
        self._stop = threading.Event()
        self._th = None
        self.e0 = self.e1 = None

    # ---- single reads (None when unsupported)
    def power_w(self):
        p, ty = C.c_uint64(0), C.c_uint32(0)
        if self.lib.rsmi_dev_power_get(self.dev, C.byref(p), C.byref(ty)) == 0:
            return p.value / 1e6
        if self.lib.rsmi_dev_current_socket_power_get(self.dev, C.byref(p)) == 0:
            return p.value / 1e6
        if self.lib.rsmi_dev_power_ave_get(self.dev, C.c_uint32(0), C.byref(p)) == 0:
            return p.value / 1e6
        return None

    def cap_w(self):
        p = C.c_uint64(0)
        return p.value / 1e6 if self.lib.rsmi_dev_power_cap_get(self.dev, C.c_uint32(0), C.byref(p)) == 0 else None

    def energy_j(self):
        e, res, ts = C.c_uint64(0), C.c_float(0), C.c_uint64(0)
        if self.lib.rsmi_dev_energy_count_get(self.dev, C.byref(e), C.byref(res), C.byref(ts)) == 0:
            return e.value * float(res.value) * 1e-6          # counter x resolution = micro-joules
        return None

    def sclk_mhz(self):
        f = _Freqs()
        if self.lib.rsmi_dev_gpu_clk_freq_get(self.dev, C.c_uint32(0), C.byref(f)) == 0 and f.current < 33:
            return f.frequency[f.current] / 1e6
        return None

    # ---- sampling window
    def _run(self):
        while not self._stop.is_set():
            w = self.power_w()
            if w is not None:
                self.watts.append(w)
                self.t.append(time.perf_counter())
            s = self.sclk_mhz()
            if s is not None:
                self.sclk.append(s)
            time.sleep(self.period)

    def __enter__(self):
        self.e0, self.t0 = self.energy_j(), time.perf_counter()
        self._th = threading.Thread(target=self._run, daemon=True)
        self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._th.join()
        self.e1, self.t1 = self.energy_j(), time.perf_counter()
        return False

    def summary(self):
        import numpy as np

        def stats(v):
            if not v:
                return None
            a = np.asarray(v, dtype=np.float64)
            return {"median": round(float(np.median(a)), 1), "p10": round(float(np.percentile(a, 10)), 1),
                    "p90": round(float(np.percentile(a, 90)), 1), "max": round(float(a.max()), 1)}
        win = self.t1 - self.t0
        ej = (self.e1 - self.e0) if (self.e0 is not None and self.e1 is not None) else None
        return {"watts": stats(self.watts), "cap_watts": self.cap_w(), "energy_j": round(ej, 1) if ej is not None else None,
                "mean_watts_from_energy": round(ej / win, 1) if ej else None, "sclk_mhz_reported": stats(self.sclk),
                "samples": len(self.watts), "window_s": round(win, 3)}


if __name__ == "__main__":                     # probe: what does this box's SMI support?  (idle reading, then 2 s beside a GEMM loop)
    import json
    import torch
    pm = PowerMeter()
    print(json.dumps({"idle": {"power_w": pm.power_w(), "cap_w": pm.cap_w(), "energy_j": pm.energy_j(), "sclk_mhz": pm.sclk_mhz()}}))
    a = torch.randn(8192, 8192, device="cuda").bfloat16()
    b = torch.randn(8192, 8192, device="cuda").bfloat16()
    torch.cuda.synchronize()
    with PowerMeter() as pm2:
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 2.0:
            for _ in range(20):
                a @ b
            torch.cuda.synchronize()
            n += 20
    s = pm2.summary()
    s["gemm_tflops"] = round(2 * 8192 ** 3 * n / s["window_s"] / 1e12, 1)
    print(json.dumps({"under_gemm_loop": s}))

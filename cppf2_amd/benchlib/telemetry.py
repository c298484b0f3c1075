"""Power and clock telemetry of the benched GPU, sampled by a CHILD PROCESS so that the timed loops never share the interpreter
lock with the sampler: `python -m cppf2_amd.benchlib.telemetry --sample` is started BEFORE bench.py initialises the GPU (a fresh
child, never a re-exec; it touches sysfs only -- no HIP, no profiler), is told the card's PCI address once the parent knows it,
samples the amdgpu hwmon files of that card (socket power, shader clock; + the driver's throttle accumulators through amdsmi where
that imports) at ~100 Hz with CLOCK_MONOTONIC time stamps, and hands everything back on `stop`.  The parent brackets its loops with
window(name) -- time.monotonic() on both sides of a device synchronisation -- and summarises the samples inside each window.

What it is for: the tuple MLP's `roofline.frac` explanation "the kernels sit at the chip's power limit" was
inferred from MFMA-busy x clock being constant across variants; with this the bench line carries the socket power against the cap
and the shader clock during the headline loops AND during each of the three MLP launch forms run back to back on their own."""
import glob
import json
import os
import select
import subprocess
import sys
import time


# ---------------------------------------------------------------------------------------------------------------------------
# child
# ---------------------------------------------------------------------------------------------------------------------------
def find_hwmon(bdf):
    """hwmon directory of the amdgpu device at PCI address `bdf` ('0000:05:00.0'), or None."""
    for dev in glob.glob("/sys/class/drm/card*/device"):
        try:
            if os.path.basename(os.path.realpath(dev)).lower() != bdf.lower():
                continue
        except OSError:
            continue
        for h in sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*"))):
            return h
    return None


def _read_int(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return None


class _Smi:
    """The driver's throttle-residency accumulators (amdsmi_get_violation_status), where the amdsmi Python package works."""

    KEYS = ("acc_ppt_pwr", "acc_socket_thrm", "acc_vr_thrm", "acc_hbm_thrm", "acc_prochot_thrm", "acc_counter",
            "per_ppt_pwr", "per_socket_thrm", "active_ppt_pwr", "active_socket_thrm")

    def __init__(self, bdf):
        self.h, self.m, self.err = None, None, None
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            self.m = amdsmi
            self.h = amdsmi.amdsmi_get_processor_handle_from_bdf(bdf)
        except Exception as e:      # noqa: BLE001  (no package, no permission, no such call: telemetry falls back to hwmon only)
            self.err = "%s: %s" % (type(e).__name__, e)

    def violation(self):
        if self.h is None:
            return None
        try:
            v = self.m.amdsmi_get_violation_status(self.h)
            return {k: v.get(k) for k in self.KEYS if isinstance(v.get(k), (int, float))}
        except Exception as e:      # noqa: BLE001
            self.err = "%s: %s" % (type(e).__name__, e)
            self.h = None
            return None

    def clocks(self):
        """Current shader clock of every XCD (MHz) from the gpu_metrics table."""
        if self.h is None:
            return None
        try:
            g = self.m.amdsmi_get_gpu_metrics_info(self.h)
            c = [x for x in (g.get("current_gfxclks") or []) if isinstance(x, (int, float)) and 0 < x < 60000]
            return c or None
        except Exception:      # noqa: BLE001
            return None


def sample_main(hz=100.0):
    """Child: wait for 'card <bdf>', sample until 'stop', print one JSON object."""
    out = {"samples": [], "smi": [], "bdf": None, "hwmon": None, "error": None}
    bdf = None
    while bdf is None:
        line = sys.stdin.readline()
        if not line or line.strip() == "stop":
            print(json.dumps(out))
            return 0
        if line.startswith("card "):
            bdf = line.split()[1]
    h = find_hwmon(bdf)
    out["bdf"], out["hwmon"] = bdf, h
    if h is None:
        out["error"] = "no amdgpu hwmon directory for " + bdf
    pfile = None
    if h is not None:
        pfile = next((os.path.join(h, n) for n in ("power1_input", "power1_average") if os.path.exists(os.path.join(h, n))), None)
        out["power_file"] = None if pfile is None else os.path.basename(pfile)
        out["power_cap_uw"] = _read_int(os.path.join(h, "power1_cap"))
        out["power_cap_max_uw"] = _read_int(os.path.join(h, "power1_cap_max"))
        labels = {}
        for n in ("freq1", "freq2"):
            try:
                with open(os.path.join(h, n + "_label")) as f:
                    labels[n] = f.read().strip()
            except OSError:
                pass
        out["freq_labels"] = labels
    smi = _Smi(bdf)
    period = 1.0 / hz
    nxt = time.monotonic()
    k = 0
    while True:
        now = time.monotonic()
        r, _, _ = select.select([sys.stdin], [], [], max(0.0, nxt - now))
        if r:
            line = sys.stdin.readline()
            if not line or line.strip() == "stop":
                break
            continue
        nxt += period
        if nxt < time.monotonic() - period:       # fell behind (a slow read): do not burst
            nxt = time.monotonic()
        if h is not None:
            t = time.monotonic()
            out["samples"].append((t, _read_int(pfile) if pfile else None, _read_int(os.path.join(h, "freq1_input"))))
        if k % 10 == 0:                             # ~10 Hz: throttle accumulators + per-XCD clocks
            v = smi.violation()
            c = smi.clocks()
            if v is not None or c is not None:
                out["smi"].append((time.monotonic(), v, c))
        k += 1
    out["smi_error"] = smi.err
    print(json.dumps(out))
    return 0


# ---------------------------------------------------------------------------------------------------------------------------
# parent
# ---------------------------------------------------------------------------------------------------------------------------
class Telemetry:
    def __init__(self):
        self.proc, self.windows, self.reason, self.bdf = None, [], None, None

    @classmethod
    def start(cls):
        """Before the parent initialises the GPU.  Never raises: without sysfs access the summary says why."""
        t = cls()
        if os.environ.get("CPPF_BENCH_NO_TELEMETRY"):
            t.reason = "CPPF_BENCH_NO_TELEMETRY"
            return t
        if not glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
            t.reason = "no amdgpu hwmon directory under /sys/class/drm"
            return t
        try:
            root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            env = {k: v for k, v in os.environ.items() if not k.startswith(("ROCP", "LD_PRELOAD", "HSA_TOOLS"))}
            t.proc = subprocess.Popen([sys.executable, "-m", "cppf2_amd.benchlib.telemetry", "--sample"], cwd=root, env=env,
                                      stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        except OSError as e:
            t.reason = "sampler not started: %s" % e
        return t

    def attach(self, props):
        """props = torch.cuda.get_device_properties(dev): tells the child which card to sample."""
        if self.proc is None:
            return
        self.bdf = "%04x:%02x:%02x.0" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
        try:
            self.proc.stdin.write("card %s\n" % self.bdf)
            self.proc.stdin.flush()
        except OSError as e:
            self.reason = "sampler gone: %s" % e
            self.proc = None

    class _Window:
        def __init__(self, tel, name, sync):
            self.tel, self.name, self.sync = tel, name, sync

        def __enter__(self):
            if self.sync:
                self.sync()
            self.t0 = time.monotonic()
            return self

        def __exit__(self, *exc):
            if self.sync:
                self.sync()
            self.tel.windows.append((self.name, self.t0, time.monotonic()))
            return False

    def window(self, name, sync=None):
        """with tel.window("loop", torch.cuda.synchronize): ...  -- the samples between the two time stamps belong to `name`."""
        return Telemetry._Window(self, name, sync)

    def finish(self, skip_s=0.05):
        """Stops the child and returns the `roofline.power` object: per window the socket power (mean / max, W), its fraction of
        the cap, the shader clock (MHz) and, where amdsmi works, the growth of the driver's power-throttle accumulator.  The first
        `skip_s` seconds of a window are left out of the means (the SMU's power figure is a moving average)."""
        if self.proc is None:
            return {"available": False, "reason": self.reason or "not started"}
        try:
            raw, _ = self.proc.communicate("stop\n", timeout=20)
            data = json.loads(raw.strip().splitlines()[-1])
        except Exception as e:      # noqa: BLE001
            try:
                self.proc.kill()
            except OSError:
                pass
            return {"available": False, "reason": "sampler failed: %s: %s" % (type(e).__name__, e)}
        return summarize(data, self.windows, skip_s)


def _stats(vals, scale):
    vals = sorted(v * scale for v in vals)
    n = len(vals)
    return {"mean": round(sum(vals) / n, 1), "min": round(vals[0], 1), "max": round(vals[-1], 1), "p50": round(vals[n // 2], 1)}


def summarize(data, windows, skip_s=0.05):
    """Pure function (tests/test_host_logic.py): sampler output + [(name, t0, t1)] -> the `roofline.power` object."""
    if data.get("error"):
        return {"available": False, "reason": data["error"], "bdf": data.get("bdf")}
    cap = data.get("power_cap_uw")
    out = {"available": True, "bdf": data.get("bdf"), "source": "amdgpu hwmon (%s, freq1_input = %s), sampled by a child process"
           % (data.get("power_file"), (data.get("freq_labels") or {}).get("freq1", "sclk")),
           "power_cap_w": None if not cap else round(cap / 1e6, 1),
           "power_cap_max_w": None if not data.get("power_cap_max_uw") else round(data["power_cap_max_uw"] / 1e6, 1),
           "samples": len(data.get("samples", [])), "smi": ("amdsmi violation status + gpu_metrics at ~10 Hz" if data.get("smi")
                                                          else "unavailable: %s" % data.get("smi_error")),
           "windows": {}}
    for name, t0, t1 in windows:
        inside = [s for s in data.get("samples", []) if t0 + min(skip_s, 0.25 * (t1 - t0)) <= s[0] <= t1]
        w = {"seconds": round(t1 - t0, 3), "samples": len(inside)}
        p = [s[1] for s in inside if s[1] is not None]
        f = [s[2] for s in inside if s[2] is not None]
        if p:
            w["power_w"] = _stats(p, 1e-6)
            if cap:
                w["power_frac_of_cap"] = {"mean": round(w["power_w"]["mean"] * 1e6 / cap, 4), "max": round(w["power_w"]["max"] * 1e6 / cap, 4)}
        if f:
            w["sclk_mhz"] = _stats(f, 1e-6)
        smi = [s for s in data.get("smi", []) if t0 <= s[0] <= t1]
        xcd = [c for s in smi if s[2] for c in s[2]]
        if xcd:
            w["sclk_mhz_all_xcds"] = _stats(xcd, 1.0)
        vio = [s for s in smi if s[1]]
        if len(vio) >= 2:
            a, b = vio[0][1], vio[-1][1]
            w["throttle"] = {k: (b[k] - a[k]) for k in a if k.startswith("acc_") and k in b and isinstance(a[k], (int, float))}
            w["throttle"].update({k: b[k] for k in b if k.startswith(("per_", "active_"))})
        out["windows"][name] = w
    return out


if __name__ == "__main__":
    if "--sample" in sys.argv:
        sys.exit(sample_main())
